"""Process-wide counter of "parameter values may have changed behind autograd's back".

Derived-weight caches (the concatenated q/k/v weights of grit_amd.models.common.attention.Attention.fused_weights, the two cross
query projections of ParallelAttentionLayer, captured decode graphs) key on (data_ptr, tensor._version, dtype).  That is not
enough under grit_amd.amp.Bf16Compute: the module's parameters are views into flat buffers that FlatAdam rewrites through a raw
kernel launch (grit_adam_flat) or a flat `copy_` -- neither bumps the views' version counters -- so an evaluation after some
training steps would keep using the weights of the first evaluation.  Every writer of that kind calls bump(); every cache
includes current() in its tag.
"""
_EPOCH = 0


def current():
    return _EPOCH


def bump():
    global _EPOCH
    _EPOCH += 1
    return _EPOCH
