"""CPU oracle for the keypoint-diffusion denoising hot path.

TEST INFRASTRUCTURE ONLY.  This package is a plain-PyTorch (CPU, fp32) restatement of
the reference algorithm for the path named in BASELINE.json's north_star.  It is imported
only by tests/, by __graft_entry__.smoke() and by bench.py's cpu_baseline leg -- always as
the checker, never as the thing that is measured or shipped.  The product package
(keypoint-diffusion_amd/) never imports it and has no CPU fallback.

Every function cites the reference file:line it follows (paths relative to the
reference checkout, e.g. models/dynamics.py:89-217).

Pinning: the dense sub-blocks (EGNN edge/coord/node MLPs, GVP chains, GVP layer norm, rbf,
noise schedule) are pinned by golden vectors generated from the reference's own importable
sub-modules (tests/golden/make_golden.py, fixtures under tests/golden/*.npz).  The graph
operations the reference delegates to un-vendored third-party libraries (DGL,
pytorch-cluster, pytorch-scatter; versions unpinned in readme.md:14-16) are restated from
their documented semantics in graph_ops.py; the reference holds no tests or vectors for
them, so that part of the oracle is "parity unpinned" (see DESIGN.md).
"""
