cd $GRAFT_REPO_ROOT
export RAYFINDER_AMD_LIB=$PWD/rayfinder_amd/librayfinder_amd_exp.so RF_DEBUG_COUNTERS=1
python - <<'PY' 2>&1 | grep -E "STEPKIND|rays"
import sys; sys.path.insert(0, ".")
import rayfinder_amd as rf
from rayfinder_amd import scenes
W, H, spp = 1920, 1080, 16
pt, info = scenes.atrium()
for b in (1, 2, 8):
    r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, rf.fly_camera(W, H), spp, b, rf.make_sky(), 0.25), pt.scene())
    r.reset_stats(); r.render(spp); r.synchronize(); s = r.stats()
    print("bounces", b, "rays", s["closest_rays"], s["shadow_rays"])
    r.close()
PY
