"""MI355X-native hot path of the learned point-cloud codec ikt-luh/Unified-Point-Cloud-Compression.

Layout
  csrc/ + libpcc_hip.so   hand-written HIP kernels behind the C ABI of include/pcc_hip.h
  lib.py                  ctypes binding (fails loudly when the library is missing; no CPU fallback)
  sparse.py               coordinate sets / kernel maps / functional operators
  MinkowskiEngine/        `import MinkowskiEngine as ME` compatible surface
  compressai/             `compressai.*` compatible surface
  model/                  UnifiedModel.compress()/decompress() counterpart of the reference's model/
  frames.py               frame/block sharding across GPUs (one process per GPU, RCCL gather)
"""
import sys

__all__ = ["install_shims"]


def install_shims(force=False):
    """Register the compatible surfaces under the names the reference imports, so that
    `model/transforms.py` / `model/blocks.py` of the reference load unchanged:

        import unified_point_cloud_compression_amd as upcc
        upcc.install_shims()
        import MinkowskiEngine as ME          # -> this package's surface
    """
    from . import MinkowskiEngine as _me
    from . import compressai as _ca
    for name, mod in (("MinkowskiEngine", _me), ("compressai", _ca)):
        if name in sys.modules and sys.modules[name] is not mod and not force:
            raise RuntimeError(f"a different `{name}` is already imported; pass force=True to replace it")
        sys.modules[name] = mod
        prefix = mod.__name__ + "."
        for k, v in list(sys.modules.items()):
            if k.startswith(prefix):
                sys.modules[name + "." + k[len(prefix):]] = v
    return _me, _ca
