"""TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

CPU oracles for the GRIT hot path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this package; nothing under grit_amd/ does (tests/test_layout.py enforces it).

  msda_oracle.c / msda.py   plain-C restatement of the reference CUDA MSDeformAttn op (fwd + bwd)
  torch_ref.py              float32/float64 PyTorch restatements of the fused attention cores
                            (window attention, decoder attention, MSDA via grid_sample)

Parity status: PINNED -- every oracle is checked against golden vectors generated in the build
container from the imported reference (tests/golden/make_golden.py, fixtures in tests/golden/*.npz).

Exception, stated where it applies: the PTB tokenizer of the self-critical reward (grit_amd/datasets/caption/metrics/tokenizer.py,
a product-side restatement of a third-party Java program that is absent from the reference tree and cannot run here) is checked
against tests/golden/ptb_rules_table.json -- a hand-checked TABLE of published PTB / CoreNLP conventions, not output of the jar:
"parity unpinned" against the Java program itself.
"""
