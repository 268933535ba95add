import time, sys, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import __graft_entry__ as g
pkg=g.load_package(); api=pkg.api
import util
hip=pkg.hip_backend("cuda:0")
D=api.Denoiser
w,h=3840,2160
dens=[D.REBLUR_DIFFUSE_SPECULAR]
scene=pkg.synth.Scene(w,h,dolly=0.01,device="cuda:0")
st=util.default_settings(api,scene,dens,minMaterialForDiffuse=0,minMaterialForSpecular=1)
hz=pkg.harness.Harness(hip,dens,w,h)
fr=scene.frame(0); planes=hz.upload(fr); ids=[int(dens[0])]
for f in range(3):
    hz.frame(scene.common_settings(api,fr,f,reset=(f==0)),planes,st)
torch.cuda.synchronize()
N=8
t=time.perf_counter()
for f in range(N):
    cs=scene.common_settings(api,fr,f+3)
t_cs=(time.perf_counter()-t)/N
t=time.perf_counter()
for f in range(N):
    hz.nrd.new_frame(); hz.nrd.set_common_settings(cs); hz.bind(planes); hz.nrd.set_denoiser_settings(ids[0],st[dens[0]])
t_bind=(time.perf_counter()-t)/N
t=time.perf_counter()
for f in range(N):
    d=hz.nrd.dispatches(ids)
t_disp=(time.perf_counter()-t)/N
torch.cuda.synchronize()
t=time.perf_counter()
for f in range(N):
    for i in range(7):
        hz.nrd.denoise_range(ids,i,1)
t_range=(time.perf_counter()-t)/N
torch.cuda.synchronize()
t=time.perf_counter()
for f in range(N):
    for i in range(7):
        hz.nrd.denoise_rows(ids,i,0,80,part=1); hz.nrd.denoise_rows(ids,i,h-80,80,part=0); hz.nrd.denoise_rows(ids,i,80,h-160,part=2)
t_rows=(time.perf_counter()-t)/N
torch.cuda.synchronize()
print("per frame host ms: common_settings %.3f  bind+settings %.3f  dispatches() %.3f  7x denoise_range %.3f (enqueue only)  21x denoise_rows %.3f"%(t_cs*1e3,t_bind*1e3,t_disp*1e3,t_range*1e3,t_rows*1e3))
