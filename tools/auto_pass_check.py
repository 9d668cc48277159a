import sys, time
sys.path.insert(0, "/root/repo")
from raytracer_amd import api
b = api.Builder(); b.register_reference_textures()
info = b.named_scene("final_scene")
ctx = api.Context(0); sc = ctx.upload(b.compile(info.world))
for spp in (256, 2048):
    p = api.params_from_info(info)          # spp_per_pass = 0: the library chooses
    st = api.rt_stats(); t = time.time()
    ctx.render(sc, info.camera, p, spp, stats=st)
    print("final_scene 800x800, %d spp, pass size left to the library: %d passes, %.2f G rays/s (%.2f s)" % (spp, st.passes, st.ray_segments / st.seconds_render / 1e9, time.time() - t))
