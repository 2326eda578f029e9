"""Autograd-facing wrappers of the gfx950 kernels (one module per kernel family)."""
