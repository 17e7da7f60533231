"""CPU oracle for the sparse-voxel codec hot path.  TEST INFRASTRUCTURE ONLY.

This package is a plain numpy restatement of the algorithms the reference
(ikt-luh/Unified-Point-Cloud-Compression) reaches through MinkowskiEngine and
CompressAI on its hot path.  It exists so
the hand-written HIP path can be checked against something independent.

PARITY UNPINNED: the reference ships no tests, golden vectors, fixtures or
weights for this path, and MinkowskiEngine / CompressAI (which hold the
arithmetic) are neither vendored under /root/reference nor installed here.  The
oracle is therefore pinned by (i) dense `torch.nn.functional.conv3d` /
`conv_transpose3d` equivalence tests in tests/test_oracle_dense.py and (ii)
closed-form checks of the entropy-model formulas, not by reference outputs.

Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import this package.  The product path
(`unified_point_cloud_compression_amd`) never does and fails loudly when the
HIP library is missing.
"""
from . import coords, ops, entropy, codec  # noqa: F401
