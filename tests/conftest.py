import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
DUCK = os.path.join(GOLDEN, "Duck.glb")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "perf: timing assertions on a real MI355X (python -m pytest tests -m perf, no -x; never part of the parity runs)")


def pytest_collection_modifyitems(config, items):
    # `-m gpu` tests need a device; everything else must pass on a CPU-only box.
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


class DuckOracle:
    """Duck.glb through the ORACLE's ingest + builder (numpy arrays)."""

    def __init__(self):
        from oracle import gltf_ref, orc
        self.model = gltf_ref.load_model(DUCK)
        self.P, self.N, self.T, self.I = gltf_ref.flatten(self.model)
        self.nodes, self.idx, self.depth = orc.build_bvh(self.P)
        self.tris36 = orc.reorder(self.P, self.idx)
        self.pos48, self.attr80 = gltf_ref.gpu_layout(self.tris36, orc.reorder(self.N, self.idx), orc.reorder(self.T, self.idx), orc.reorder(self.I, self.idx))
        self.descs, self.texels = gltf_ref.flatten_textures(self.model["textures"])
        self.scene = orc.OracleScene(self.nodes, self.pos48, self.attr80, self.descs, self.texels)


@pytest.fixture(scope="session")
def duck_oracle():
    return DuckOracle()


@pytest.fixture(scope="session")
def duck_pt():
    import rayfinder_amd as rf
    return rf.PtFormat.from_gltf(DUCK)


def oracle_scene_from_pt(pt):
    """OracleScene over a product PtFormat's arrays (for GPU-vs-oracle parity on any scene)."""
    from oracle import orc
    a = pt.arrays()
    descs, off = [], 0
    for (px, w, h) in a["baseColorTextures"]:
        descs.append((w, h, off))
        off += px.size
    texels = np.concatenate([px for (px, _, _) in a["baseColorTextures"]]) if descs else np.array([0xFFFFFFFF], np.uint32)
    if not descs:
        descs = [(1, 1, 0)]
    return orc.OracleScene(a["bvhNodes"], a["trianglePositionAttributes"], a["triangleVertexAttributes"], np.array(descs, np.uint32), texels), a


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)
