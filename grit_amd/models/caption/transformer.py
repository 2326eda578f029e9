"""GRIT captioner: detector -> grid network -> caption decoder; teacher forcing and beam search.

Mirror of reference models/caption/transformer.py (Transformer.__init__ :14-51, forward :53-132, step :134-169,
select :184-188, _expand_state :190-202, iter :204-254).  The beam loop keeps the reference's observable
semantics to the token: always `max_len` steps, candidates = running score + word log-prob,
a finished beam keeps its score only at vocabulary index 0 (-999 elsewhere), selection = descending torch.sort of
the flattened [beam*V] candidates, beam = index // V, every registered state is re-gathered per step, final beams
re-sorted by score.
"""
import ctypes
import os

import torch
from torch import nn

from grit_amd import lib as _lib
from grit_amd.models.caption.base import BaseCaptioner
from grit_amd.ops import backend
from grit_amd.ops import beam as beam_ops
from grit_amd.models.caption.cap_generator import CaptionGenerator
from grit_amd.models.caption.grid_net import GridFeatureNetwork
from grit_amd.utils.misc import NestedTensor


# GRIT_GRAPH_DECODE=0: never replay beam search from a captured HIP graph (A/B knob, tools/bench_decode.py)
_GRAPH_DECODE = os.environ.get('GRIT_GRAPH_DECODE', '1') != '0'
# GRIT_FUSED_BEAM_STEP=0: the candidate / selection / gather arithmetic of a beam step as separate torch ops + grit_topk_rows_f32
_FUSED_BEAM_STEP = os.environ.get('GRIT_FUSED_BEAM_STEP', '1') != '0'


class Transformer(BaseCaptioner):

    def __init__(self, detector, config=None):
        super().__init__()
        self._decode_graphs = {}
        m = config.model
        # d_model 512 / 8 heads / d_ff 2048 are the constructor defaults on purpose: the reference never forwards
        # config.model.n_heads, grid_net.n_memories or cap_generator.decoder_name
        self.grid_net = GridFeatureNetwork(n_layers=m.grid_net.n_layers, d_in=m.grid_feat_dim, dropout=m.dropout)
        self.cap_generator = CaptionGenerator(n_layers=m.cap_generator.n_layers, vocab_size=m.vocab_size,
                                              max_len=m.max_len, pad_idx=m.pad_idx, dropout=m.dropout,
                                              cfg=m.cap_generator)
        self.config = config
        self.bos_idx = m.bos_idx
        self.use_reg_feat, self.use_gri_feat = m.use_reg_feat, m.use_gri_feat
        self.cached_features = False
        if self.use_gri_feat:
            self.register_state('gri_feat', None)
            self.register_state('gri_mask', None)
        if self.use_reg_feat:
            self.register_state('reg_feat', None)
            self.register_state('reg_mask', None)
        self.init_weights()
        self.detector = detector  # attached after init_weights: the detector keeps its own initialisation

    def init_weights(self):
        for p in self.parameters():
            if p.dim() > 1:
                nn.init.xavier_uniform_(p)

    # ------------------------------------------------------------------ visual side
    def _visual_inputs(self, images):
        vis = images if self.cached_features else self.detector(images)
        vis = dict(vis)
        if self.config.model.use_gri_feat:
            if len(self.grid_net.layers) > 0 and type(self.grid_net).forward is GridFeatureNetwork.forward:
                vis['gri_feat'] = self.grid_net.last(vis['gri_feat'], vis['gri_mask'])  # == grid_net(...)[0][:, -1]
            else:
                grid, _ = self.grid_net(vis['gri_feat'], vis['gri_mask'])
                vis['gri_feat'] = grid[:, -1]
        return vis

    def forward(self, images, seq, use_beam_search=False, max_len=20, eos_idx=3, beam_size=5, out_size=1,
                return_probs=False, **kwargs):
        if not use_beam_search:
            return self.cap_generator(seq, self._visual_inputs(images))
        return self.beam_search(images, max_len, eos_idx, beam_size, out_size, return_probs, **kwargs)

    # ------------------------------------------------------------------ beam search
    def beam_search(self, images, max_len, eos_idx, beam_size, out_size, return_probs, **kwargs):
        """Inference on a HIP device (no autograd, eval mode): the decode loop -- grid net + `max_len` decoder steps + beam
        bookkeeping, ~5 000 small kernels whose launches bound the loop at ~2.7 ms of host time per step -- is captured ONCE per
        (batch, beam, length, feature shapes) into a HIP graph and replayed from then on (next-row N1: the device-side beam
        loop).  Same kernels, same order, same data: tokens and log-probs are those of the eager loop bit for bit.  Training
        (self-critical: beam search with gradient), CPU runs and return_probs keep the eager loop."""
        if self._graph_eligible(images, return_probs, kwargs):
            vis = images if self.cached_features else self.detector(images)
            return self._beam_search_graphed(dict(vis), max_len, eos_idx, beam_size, out_size)
        from grit_amd.ops.linear import suspend_single_use
        with suspend_single_use():  # with gradient (self-critical training) the decoder's weights are used once per STEP
            return self._beam_search_eager(images, max_len, eos_idx, beam_size, out_size, return_probs, **kwargs)

    def _graph_eligible(self, images, return_probs, kwargs):
        if not _GRAPH_DECODE or return_probs or kwargs or self.training or torch.is_grad_enabled():
            return False
        _, device = self.get_bs_device(images)
        return device.type == 'cuda' and not torch.cuda.is_current_stream_capturing()

    def _decode_from_features(self, vis, max_len, eos_idx, beam_size, out_size):
        was = self.cached_features
        self.cached_features = True
        try:
            return self._beam_search_eager(vis, max_len, eos_idx, beam_size, out_size, False)
        finally:
            self.cached_features = was

    def _beam_search_graphed(self, vis, max_len, eos_idx, beam_size, out_size):
        names = sorted(k for k, v in vis.items() if isinstance(v, torch.Tensor))
        # The graph reads every parameter from its storage at replay time (derived weights are rebuilt inside the capture), so
        # in-place writers -- optimizer steps, load_state_dict -- need no invalidation.  What does: parameter STORAGE being
        # replaced (module.to(dtype), Bf16Compute wrapping): the addresses of the first and last decoder parameters are in the key.
        probe = (self.grid_net.fc.weight, self.cap_generator.fc.weight)
        key = (max_len, eos_idx, beam_size, out_size, torch.is_inference_mode_enabled(), vis[names[0]].device.index) + \
            tuple((p.data_ptr(), p.dtype) for p in probe) + \
            tuple((k, tuple(vis[k].shape), vis[k].dtype) for k in names)
        entry = self._decode_graphs.pop(key, None)
        if entry is not None:
            self._decode_graphs[key] = entry  # most recently used last
        if entry is None:
            while len(self._decode_graphs) >= 8:  # a few live shapes at most (each graph keeps its activations): drop the
                self._decode_graphs.pop(next(iter(self._decode_graphs)))  # least recently used one, not all of them
            static_in = {k: vis[k].clone() for k in names}
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):  # warm-up on a side stream: library workspaces, lazily built caches
                self._decode_from_features(dict(static_in), max_len, eos_idx, beam_size, out_size)
            cur.wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = self._decode_from_features(dict(static_in), max_len, eos_idx, beam_size, out_size)
            entry = self._decode_graphs[key] = (graph, static_in, static_out)
        graph, static_in, static_out = entry
        for k in names:
            if vis[k].data_ptr() != static_in[k].data_ptr():
                static_in[k].copy_(vis[k])
        graph.replay()
        return tuple(o.clone() for o in static_out)

    def _beam_search_eager(self, images, max_len, eos_idx, beam_size, out_size, return_probs, **kwargs):
        batch_size, device = self.get_bs_device(images)
        self.seq_mask = torch.ones((batch_size, beam_size, 1), device=device)      # 1 while the beam is alive
        self.seq_logprob = torch.zeros((batch_size, 1, 1), device=device)          # running score per beam
        self.log_probs, self.selected_words = [], None
        if return_probs:
            self.all_log_probs = []
        outputs = []
        cross = [m.attention for layer in self.cap_generator.layers for m in (layer.vis_att1, layer.vis_att2)]
        try:
            for att in cross:  # N1: project the visual memory once per beam layout instead of once per step
                att.hoist_kv, att._kv = True, None
            for m in self._kv_cached_modules():
                m._kv_deferred, m._beam_src = False, None
            with self.statefulness(batch_size):
                for t in range(max_len):
                    images, outputs = self.iter(timestep=t, samples=images, outputs=outputs, return_probs=return_probs,
                                                batch_size=batch_size, beam_size=beam_size, eos_idx=eos_idx, **kwargs)
        finally:
            for att in cross:
                att.hoist_kv, att._kv = False, None
        _, order = torch.sort(self.seq_logprob, 1, descending=True)
        idx = order.expand(batch_size, beam_size, max_len)
        outputs = torch.gather(torch.cat(outputs, -1), 1, idx)
        log_probs = torch.gather(torch.cat(self.log_probs, -1), 1, idx)
        outputs, log_probs = outputs.contiguous()[:, :out_size], log_probs.contiguous()[:, :out_size]
        if out_size == 1:
            outputs, log_probs = outputs.squeeze(1), log_probs.squeeze(1)
        if not return_probs:
            return outputs, log_probs
        all_lp = torch.cat(self.all_log_probs, 2)
        all_lp = torch.gather(all_lp, 1, order.unsqueeze(-1).expand(batch_size, beam_size, max_len, all_lp.shape[-1]))
        return outputs, log_probs, all_lp

    def step(self, timestep, prev_output, samples, seq, mode='teacher_forcing', **kwargs):
        if mode != 'feedback':
            raise NotImplementedError
        if timestep == 0:
            vis = samples if self.cached_features else self.detector(samples)
            if self.config.model.use_gri_feat:
                grid, self.gri_mask = self.grid_net(vis['gri_feat'], vis['gri_mask'])
                self.gri_feat = grid[:, -1]
            if self.config.model.use_reg_feat:
                self.reg_feat, self.reg_mask = vis['reg_feat'], vis['reg_mask']
            feat = getattr(self, 'gri_feat', self.reg_feat)
            it = feat.data.new_full((feat.shape[0], 1), self.bos_idx).long()
        else:
            it = prev_output
        vis_inputs = {}
        if self.config.model.use_gri_feat:
            vis_inputs.update(gri_feat=self.gri_feat, gri_mask=self.gri_mask)
        if self.config.model.use_reg_feat:
            vis_inputs.update(reg_feat=self.reg_feat, reg_mask=self.reg_mask)
        return self.cap_generator(it, vis_inputs)

    def get_bs_device(self, samples):
        if isinstance(samples, dict):
            t = samples['gri_feat' if 'gri_feat' in samples else 'reg_feat']
        elif isinstance(samples, NestedTensor):
            t = samples.tensors
        else:
            raise TypeError("images must be a NestedTensor or a dict of cached features")
        return t.shape[0], t.device

    def init_state(self, batch_size, device):
        return [torch.zeros((batch_size, 0), dtype=torch.long, device=device), None, None]

    def select(self, t, candidate_logprob, beam_size, **kwargs):
        """[B, Beam, V] -> the `beam_size` best of the flattened candidates, best first.  The reference takes the head of
        a full descending torch.sort over beam*V = 51 005 scores per image per step (transformer.py:184-188); the selection
        kernel (grit_topk_rows_f32: one workgroup per image, the row read once) returns the same values in the same order --
        the order among exactly tied scores is unspecified in the reference and is by ascending index here."""
        flat = candidate_logprob.reshape(candidate_logprob.shape[0], -1)
        if flat.is_cuda and flat.dtype == torch.float32 and beam_size <= 8 and not flat.requires_grad \
                and backend.override() is None:
            flat = flat if flat.stride(1) == 1 else flat.contiguous()
            idx = torch.empty((flat.shape[0], beam_size), dtype=torch.int64, device=flat.device)
            logprob = torch.empty((flat.shape[0], beam_size), dtype=torch.float32, device=flat.device)
            with _lib.device_guard(flat.device):
                st = _lib.load().grit_topk_rows_f32(ctypes.c_void_p(flat.data_ptr()), flat.stride(0), flat.shape[0],
                                                    flat.shape[1], beam_size, ctypes.c_void_p(idx.data_ptr()),
                                                    ctypes.c_void_p(logprob.data_ptr()), _lib.current_stream_ptr())
            _lib.check(st, "grit_topk_rows_f32")
            return idx, logprob
        # with gradient (self-critical training), on the CPU (oracle runs) and for wide beams: the library selection
        logprob, idx = torch.topk(flat, beam_size, dim=-1, largest=True, sorted=True)
        return idx, logprob

    def _expand_state(self, selected_beam, cur_beam_size, batch_size, beam_size):
        """state [B*cur_beam, ...] -> rows of the surviving beams [B*beam, ...]."""

        def fn(tensor):
            tail = [int(s) for s in tensor.shape[1:]]
            index = selected_beam.view(batch_size, beam_size, *([1] * len(tail))).expand(batch_size, beam_size, *tail)
            picked = torch.gather(tensor.view(batch_size, cur_beam_size, *tail), 1, index)
            return picked.view(-1, *tail)

        return fn

    def _kv_cached_modules(self):
        return [layer.self_att for layer in self.cap_generator.layers]

    def iter(self, timestep, samples, outputs, return_probs, batch_size, beam_size=5, eos_idx=3, **kwargs):
        cur_beam = 1 if timestep == 0 else beam_size
        word_logprob = self.step(timestep, self.selected_words, samples, None, mode='feedback', **kwargs)
        word_logprob = word_logprob.view(batch_size, cur_beam, -1)
        V = word_logprob.shape[-1]
        fused = _FUSED_BEAM_STEP and not return_probs and not kwargs and beam_ops.supported(word_logprob, cur_beam, beam_size)
        if fused:
            # masking, candidate scores, selection and the score / mask / log-prob gathers in two launches (ops/beam.py)
            selected_beam, selected_words, seq_logprob, seq_mask, picked = beam_ops.beam_step(
                word_logprob, self.seq_logprob, self.seq_mask if timestep > 0 else None,
                self.selected_words if timestep > 0 else None, eos_idx, beam_size)
        else:
            candidates = self.seq_logprob + word_logprob
            if timestep > 0:
                alive = (self.selected_words.view(batch_size, cur_beam) != eos_idx).float().unsqueeze(-1)
                self.seq_mask = self.seq_mask * alive
                word_logprob = word_logprob * self.seq_mask
                frozen = self.seq_logprob.expand_as(candidates).contiguous()
                frozen[:, :, 1:] = -999  # a finished beam survives only through vocabulary index 0
                candidates = self.seq_mask * candidates + frozen * (1 - self.seq_mask)

            selected_idx, selected_logprob = self.select(timestep, candidates, beam_size, **kwargs)
            selected_beam = torch.div(selected_idx, V, rounding_mode='floor')
            selected_words = selected_idx - selected_beam * V

        expand = self._expand_state(selected_beam, cur_beam, batch_size, beam_size)
        # All beams of an image carry the SAME visual memory: re-gathering it by the surviving beam index (reference :229)
        # moves nothing, and replicating it per beam at step 0 only multiplies what every cross-attention reads.  The four
        # visual states stay [B, ...]; Attention.forward groups the beams of an image onto its one copy.  Leaving them
        # untouched also keeps their identity, which is what the hoisted K/V projections are keyed on.
        visual = {id(getattr(self, n, None)) for n in ('gri_feat', 'gri_mask', 'reg_feat', 'reg_mask')} - {id(None)}
        # Self-attention caches that update themselves (one launch: source beam's history + the new key / value, ops/kv_cache.py)
        # only need to be told which beam each survivor continues; everything else is re-gathered as in the reference.
        for m in self._kv_cached_modules():
            if m._kv_deferred:
                m._beam_src = selected_beam
                visual.update((id(m.running_keys), id(m.running_values)))
        self.apply_to_states(lambda tensor: tensor if id(tensor) in visual else expand(tensor))

        beam_col = selected_beam.unsqueeze(-1)
        if fused:
            self.seq_logprob, self.seq_mask = seq_logprob, seq_mask
        else:
            self.seq_logprob = selected_logprob.unsqueeze(-1)
            self.seq_mask = torch.gather(self.seq_mask, 1, beam_col)
        # the history of every surviving beam moves with it: ONE gather of the concatenated columns instead of one per past
        # step (the lists hold a single [B, beam, t] tensor from step 1 on)
        if outputs:
            hist = outputs[0] if len(outputs) == 1 else torch.cat(outputs, -1)
            outputs = [torch.gather(hist, 1, beam_col.expand(batch_size, beam_size, hist.shape[-1]))]
        outputs.append(selected_words.unsqueeze(-1))
        if return_probs:
            lp = word_logprob.expand((batch_size, beam_size, -1)) if timestep == 0 else word_logprob
            self.all_log_probs.append(lp.unsqueeze(2))
        if not fused:
            picked = torch.gather(word_logprob, 1, beam_col.expand(batch_size, beam_size, V))
            picked = torch.gather(picked, 2, selected_words.unsqueeze(-1))
        if self.log_probs:
            hist = self.log_probs[0] if len(self.log_probs) == 1 else torch.cat(self.log_probs, -1)
            self.log_probs = [torch.gather(hist, 1, beam_col.expand(batch_size, beam_size, hist.shape[-1]))]
        self.log_probs.append(picked)
        self.selected_words = selected_words.view(-1, 1)
        return samples, outputs
