import sys; sys.path.insert(0,'.')
import torch
from gltf_renderer_amd import scenes, abi
from gltf_renderer_amd.renderer import Renderer
s=scenes.sponza_class(); r=Renderer(device=0); h=s.upload(r); r.build_accel()
out=r.create_output(s.width,s.height); r.enable_counters(True); r.reset_stats()
for f in range(4): r.trace(s.settings, s.execute_params(frame=f, env_handle=h["env"]), out)
torch.cuda.synchronize(); c=r.stats()
print({k:getattr(c,k) for k in dir(c) if not k.startswith('_') and isinstance(getattr(c,k),(int,float))})
