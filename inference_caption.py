"""Single-image captioning (reference inference_caption.py:34-69): build detector + Transformer, optional checkpoint
(`state_dict` key, strict=False), NestedTensor batch of one, beam search, token ids -> words through vocab.json.

    python inference_caption.py --img image.npy|image.png [--checkpoint ckpt.pth] [--vocab data/vocab.json] [--beam 5]

`caption_tokens()` is the plumbing the tests drive (BASELINE config 1: 224x224, beam 1 = greedy, CPU with the
oracle ops injected)."""
import argparse
import json
import os

import torch

from grit_amd.config import default_config
from grit_amd.tuning import load_tuned_gemms
from models.caption import Transformer
from models.caption.detector import build_detector
from engine.utils import nested_tensor_from_tensor_list


def build_model(config, device, checkpoint=''):
    if torch.device(device).type == 'cuda':
        load_tuned_gemms()  # library GEMM solutions tuned for this model's shapes (grit_amd/tuning.py)
    detector = build_detector(config).to(device)
    model = Transformer(detector=detector, config=config).to(device)
    if checkpoint and os.path.exists(checkpoint):
        state = torch.load(checkpoint, map_location='cpu')
        missing, unexpected = model.load_state_dict(state['state_dict'], strict=False)
        print(f"model missing:{len(missing)} model unexpected:{len(unexpected)}")
    model.cached_features = False
    return model.eval()


@torch.inference_mode()
def caption_tokens(model, image, config, beam_size=None):
    """image [3,H,W] (already normalised) -> (tokens [1, beam_len] int64, log_probs [1, beam_len])."""
    device = next(model.parameters()).device
    images = nested_tensor_from_tensor_list([image]).to(device)
    return model(images, seq=None, use_beam_search=True, max_len=config.model.beam_len, eos_idx=config.model.eos_idx,
                 beam_size=beam_size or config.model.beam_size, out_size=1, return_probs=False)


@torch.inference_mode()
def caption_stream(model, batches, config, beam_size=None):
    """Caption a sequence of batches (NestedTensor each), yielding (tokens, log_probs) per batch in order -- the inner loop
    of evaluate_metrics (reference engine/caption_engine.py:156-175) arranged for the GPU: the detector of batch i+1 is
    enqueued on its own HIP stream BEFORE the beam search of batch i.  The beam-search loop is launch bound (~60 ms of
    host time for < 20 ms of GPU work at batch 64) while the detector is GPU bound (~48 ms), so the two overlap almost
    entirely.  Same kernels on the same data: the results are those of the sequential loop."""
    beam = beam_size or config.model.beam_size
    device = next(model.parameters()).device
    if device.type != 'cuda':
        for samples in batches:
            yield model(samples, seq=None, use_beam_search=True, max_len=config.model.beam_len, eos_idx=config.model.eos_idx,
                        beam_size=beam, out_size=1, return_probs=False)
        return
    # The tuned GEMM table (grit_amd/tuning.py) is switched off while two streams issue library GEMMs side by side: with it, the
    # detector stream stalls for tens of ms every few batches next to the replayed decode graph (63-91 ms per batch instead of 51,
    # tools/micro/dbg_pipelined.py); the sequential path keeps it (detector 46.6 -> 40.5 ms).
    import torch.cuda.tunable as tunable
    tuned = tunable.is_enabled()
    tunable.enable(False)
    try:
        yield from _caption_stream_device(model, batches, config, beam, device)
    finally:
        tunable.enable(tuned)


def _caption_stream_device(model, batches, config, beam, device):
    det_stream, dec_stream = torch.cuda.Stream(device), torch.cuda.Stream(device)
    caller = torch.cuda.current_stream(device)
    det_stream.wait_stream(caller)
    dec_stream.wait_stream(caller)

    def beam_search(vis, ready):
        was = model.cached_features
        model.cached_features = True
        try:
            with torch.cuda.stream(dec_stream):
                dec_stream.wait_event(ready)
                for v in vis.values():
                    v.record_stream(dec_stream)
                out = model(vis, seq=None, use_beam_search=True, max_len=config.model.beam_len,
                            eos_idx=config.model.eos_idx, beam_size=beam, out_size=1, return_probs=False)
                done = torch.cuda.Event()
                done.record(dec_stream)
        finally:
            model.cached_features = was
        caller.wait_event(done)
        for o in out:
            o.record_stream(caller)
        return out

    pending = None
    for samples in batches:
        # `batches` may produce its items lazily on the caller's stream (device collator kernels, pinned-memory uploads):
        # order the detector after whatever has been enqueued there so far, and keep the batch's memory from being reused by
        # the caller-side allocator while the detector stream still reads it
        fetched = torch.cuda.Event()
        fetched.record(caller)
        det_stream.wait_event(fetched)
        for tns in (getattr(samples, 'tensors', None), getattr(samples, 'mask', None)):
            if isinstance(tns, torch.Tensor) and tns.is_cuda:
                tns.record_stream(det_stream)
        with torch.cuda.stream(det_stream):
            vis = dict(model.detector(samples))
            ready = torch.cuda.Event()
            ready.record(det_stream)
        if pending is not None:
            yield beam_search(*pending)
        pending = (vis, ready)
    if pending is not None:
        yield beam_search(*pending)


def decode(tokens, vocab_path, eos_idx=3):
    """ids -> words, cut at the first <eos> (reference datasets/caption/field.py:258-283)."""
    with open(vocab_path) as f:
        vocab = json.load(f)
    itos = vocab['itos'] if isinstance(vocab, dict) and 'itos' in vocab else vocab
    out = []
    for row in tokens.tolist():
        words = []
        for t in row:
            if t == eos_idx:
                break
            words.append(itos[t])
        out.append(' '.join(words))
    return out


def run_main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--img', required=True)
    ap.add_argument('--checkpoint', default='')
    ap.add_argument('--vocab', default=os.path.join(os.path.dirname(os.path.abspath(__file__)), 'data', 'vocab.json'))
    ap.add_argument('--beam', type=int, default=None)
    a = ap.parse_args()
    config = default_config()
    device = torch.device('cuda:0')  # the kernels need a HIP device; there is no CPU path
    model = build_model(config, device, a.checkpoint)
    if a.img.endswith('.npy'):
        import numpy as np
        image = torch.from_numpy(np.load(a.img)).float()
    else:
        import numpy as np
        from PIL import Image
        rgb = np.asarray(Image.open(a.img).convert('RGB'), dtype=np.float32) / 255.0
        mean, std = np.array([0.485, 0.456, 0.406], np.float32), np.array([0.229, 0.224, 0.225], np.float32)
        image = torch.from_numpy(((rgb - mean) / std).transpose(2, 0, 1))
    tokens, _ = caption_tokens(model, image, config, a.beam)
    print(decode(tokens, a.vocab, config.model.eos_idx)[0] if os.path.exists(a.vocab) else tokens.tolist())


if __name__ == "__main__":
    run_main()
