"""CPU oracle for the tlxcv_amd hot path — TEST INFRASTRUCTURE, never shipped, never measured.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
Contents:
  functional.py  fp32 torch-CPU restatements of the reference forward graphs, each function citing
                 the reference file:line it follows.  torch's CPU ATen/oneDNN kernels are the same
                 arithmetic TensorLayerX's torch backend delegates to.
  tlx_cpu/       a minimal torch-CPU stand-in for the `tensorlayerx` API, used ONLY in the
                 development container by gen_golden.py to import the reference's model files
                 unmodified from /root/reference and check the restatements against them.
  gen_golden.py  writes tests/golden/*.npz (inputs recipe ids + expected logits).

PARITY PINNING: the reference ships no tests, golden vectors or fixtures, and TensorLayerX (where
the arithmetic lives, `tensorlayerx>=0.5.8`, not vendored) is not installable here.  The *graphs*
are pinned by running the reference's own model files (torch-capable ones) on tlx_cpu; the layer
*semantics* of TensorLayerX are restated from its documentation and are UNPINNED — see DESIGN.md.
"""
