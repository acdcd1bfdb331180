import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from gltf_renderer_amd import abi, scenes
from gltf_renderer_amd.renderer import Renderer
s = scenes.sponza_class(); r = Renderer(0); h = s.upload(r); r.build_accel()
out = r.create_output(s.width, s.height)
st = abi.PtSettings.from_buffer_copy(bytes(s.settings)); st.reset = 1
K = 12
for n, spp in ((8, 1), (4, 1), (1, 1), (8, 8)):
    for sb in (256, 512, 768, 1024, 1536):
        r.set_kernel_mode(abi.MODE_WAVEFRONT, sb); r.set_samples_per_trace(spp)
        for f in range(2): r.trace(st, s.execute_params(frame=f * spp, tile_rank=0, tile_rank_count=n, env_handle=h["env"]), out)
        torch.cuda.synchronize(); r.reset_stats(); t0 = time.perf_counter()
        for f in range(K): r.trace(st, s.execute_params(frame=(10 + f) * spp, tile_rank=0, tile_rank_count=n, env_handle=h["env"]), out)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / K; c = r.stats()
        print("N=%d spp=%d stage_blocks=%4d: %7.3f ms/launch %6.0f Mrays/s" % (n, spp, sb, dt * 1e3, c.rays / K / dt / 1e6))
