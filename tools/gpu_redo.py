import os, sys
os.environ["RF_DEBUG_COUNTERS"] = "1"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import rayfinder_amd as rf
from rayfinder_amd import scenes
W, H, spp, b = 1920, 1080, 8, 8
pt, info = scenes.atrium()
r = rf.ReferencePathTracer(rf.make_render_parameters(W, H, rf.fly_camera(W, H), spp, b, rf.make_sky(), 0.25), pt.scene())
r.reset_stats(); r.render(spp); r.synchronize(); r.stats()
