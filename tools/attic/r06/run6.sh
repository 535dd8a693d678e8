#!/bin/bash
OUT=gpurun_out/r06_raygen; mkdir -p $OUT
export PYTHONPATH=$PWD
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "not bench_line and not config5" > $OUT/pytest.log 2>&1; grep -E "passed|failed" $OUT/pytest.log | tail -1
python tools/r06/ab_variants.py 64 "dense_raygen=0" "dense_raygen=1" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_dense_raygen_64spp.log
python tools/r06/ab_variants.py 320 "dense_raygen=0" "dense_raygen=1" 2>&1 | grep -v amdgpu.ids | tee $OUT/ab_dense_raygen_320spp.log
RF_SCENE=duck python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06_raygen/ragged.log
import numpy as np, rayfinder_amd as rf, os
pt = rf.PtFormat.from_gltf("tests/golden/Duck.glb")
for (W, H, spp, g) in ((333, 217, 5, 0), (800, 600, 64, 0), (200, 150, 7, -1), (65, 33, 3, 0)):
    imgs = []
    for dense in (0, 1):
        r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, rf.fly_camera(W, H), spp, 4, rf.make_sky(), 0.25), pt.scene())
        r.set_option("dense_raygen", dense); r.set_option("slot_group_shift", g)
        r.render(spp); imgs.append(r.read_accumulation()[0]); st = r.stats(); r.close()
    print(W, H, spp, g, "identical" if np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32)) else "DIFFERENT", st["primary_rays"], st["closest_rays"])
PY
