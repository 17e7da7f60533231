"""In this container only: the reference's own `model/blocks.py` and `model/transforms.py` import and construct
UNCHANGED against the MinkowskiEngine / compressai surfaces of this build, and their state_dict equals the
build's counterpart key for key (north_star: "model/transforms.py and model/blocks.py load unchanged").
The reference files are read from /root/reference and never copied."""
import importlib
import sys

import pytest

pytestmark = pytest.mark.reference

CFG = {"C_in": 4, "C_out": 3, "N1": 128, "N2": 128, "N3": 128, "N4": 128}


@pytest.fixture()
def reference_modules():
    import unified_point_cloud_compression_amd as upcc
    saved = {k: v for k, v in sys.modules.items()
             if k == "model" or k.startswith("model.") or k.split(".")[0] in ("MinkowskiEngine", "compressai")}
    for k in saved:
        del sys.modules[k]
    upcc.install_shims(force=True)
    sys.path.insert(0, "/root/reference")
    try:
        blocks = importlib.import_module("model.blocks")
        transforms = importlib.import_module("model.transforms")
        assert blocks.__file__.startswith("/root/reference/") and transforms.__file__.startswith("/root/reference/")
        yield blocks, transforms
    finally:
        sys.path.remove("/root/reference")
        for k in list(sys.modules):
            if k == "model" or k.startswith("model.") or k.split(".")[0] in ("MinkowskiEngine", "compressai"):
                del sys.modules[k]
        sys.modules.update(saved)


def test_reference_transforms_construct_on_the_shim(reference_modules):
    blocks, transforms = reference_modules
    from unified_point_cloud_compression_amd.model import transforms as mine
    import unified_point_cloud_compression_amd.MinkowskiEngine as ME
    assert transforms.ME is ME
    for ref_cls, my_cls in ((transforms.AnalysisTransform, mine.AnalysisTransform),
                            (transforms.SparseSynthesisTransform, mine.SparseSynthesisTransform)):
        ref, my = ref_cls(CFG), my_cls(CFG)
        sd_r, sd_m = ref.state_dict(), my.state_dict()
        assert list(sd_r) == list(sd_m)
        for k in sd_r:
            assert sd_r[k].shape == sd_m[k].shape, k
        my.load_state_dict(sd_r)                       # the build's modules accept the reference modules' parameters
    g = blocks.MinkowskiGDN(128, inverse=True)
    assert g.inverse and tuple(g.gamma.shape) == (128, 128) and g.kernel_size == 1
    k = ref.up_1[2].kernel
    assert tuple(k.shape) == (125, 128, 128) and tuple(ref.color_conv[0].kernel.shape) == (32, 3)   # ME layout (A.4)
    assert tuple(ref.down_conv.kernel.shape) == (27, 1, 1) and ref.down_conv.bias is None


def test_reference_loss_constructs_pooling_stubs(reference_modules):
    import unified_point_cloud_compression_amd.MinkowskiEngine as ME
    # loss.py:124-125,185 only construct these
    ME.MinkowskiAvgPooling(kernel_size=3, stride=1, dimension=3)
    ME.MinkowskiChannelwiseConvolution(1, kernel_size=3, stride=1, dimension=3)
