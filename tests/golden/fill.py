"""Closed-form, name-keyed parameter fill shared by the golden generator (applied to the imported reference
modules) and the tests (applied to this repo's modules): no weight files have to travel.

Every parameter is drawn from a CPU generator seeded with crc32(name); scale depends on the role:
matrices ~ N(0, 1/fan_in) (activations stay O(1)), norm scales ~ 1 + 0.1 N, biases ~ 0.1 N, relative-position bias
tables ~ 0.5 N, sampling offsets ~ N(0,1) so that the deformable sampling points spread over the maps."""
import zlib

import torch

SKIP = ()  # nothing is skipped: Transformer.init_weights (reference transformer.py:47-51) xavier-initialises every
# >1-d parameter, the frozen sinusoid table `pos_emb` included, so its construction-time value is RNG dependent


@torch.no_grad()
def deterministic_fill_(module, prefix=''):
    for name, p in sorted(module.named_parameters()):
        if any(s in name for s in SKIP):
            continue
        g = torch.Generator().manual_seed(zlib.crc32((prefix + name).encode()) & 0x7FFFFFFF)
        r = torch.randn(p.shape, generator=g, dtype=torch.float32)
        if 'relative_position_bias_table' in name:
            r = r * 0.5
        elif name.endswith('sampling_offsets.bias'):
            r = r * 1.0
        elif 'word_emb' in name or 'pos_emb' in name or 'query_embed' in name or 'level_embed' in name:
            r = r * 0.5
        elif p.dim() >= 2:
            fan_in = p[0].numel()
            r = r / fan_in**0.5
        elif name.endswith('weight'):  # LayerNorm / GroupNorm scale
            r = 1.0 + 0.1 * r
        else:
            r = 0.1 * r
        p.copy_(r.to(p.dtype))
    return module
