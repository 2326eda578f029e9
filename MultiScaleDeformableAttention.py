"""Drop-in for the reference's compiled extension of the same import name
(models/ops/setup.py:52-60, models/ops/src/vision.cpp:13-16): `import MultiScaleDeformableAttention as MSDA`
keeps working, but the two functions now run the gfx950 kernels of libgrit_hip.so."""
from grit_amd.ops.msda import ms_deform_attn_backward, ms_deform_attn_forward  # noqa: F401
