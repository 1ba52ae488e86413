"""Monte-Carlo of one wave's search / service schedule (not a pytest; CPU only): rays of Gamma(2) length with mean 13 visits, a third of
them ending in a miss; the wave leaves the search loop when fewer than `thr` lanes search (full service pass, cost C_full steps)
or - the "light" service pass of DESIGN.md 9 - when `light_min` lanes wait with a ray that left the scene (cost C_light).
Prints search-loop occupancy and cost per ray. Calibrated to tests/tools/step_probe.py on C3 (0.60 occupied, service a quarter of
the time). Result: the light pass never pays, whatever it costs between 1.5 and 3.5 steps."""
import numpy as np
rng=np.random.default_rng(2)
def ray_len(n): return np.maximum(1,np.round(rng.gamma(2.0,13/2.,n))).astype(int)
def sim(thr=16, light_min=0, C_full=5.9, C_light=2.0, p_miss=0.34, steps_total=60000):
    rem=ray_len(64); miss=rng.random(64)<p_miss
    act_sum=0; steps=0; cost=0.0; rays=0
    while steps<steps_total:
        act=rem>0
        done=~act
        if act.sum()<thr:
            n=done.sum(); rays+=n
            rem[done]=ray_len(n); miss[done]=rng.random(n)<p_miss
            cost+=C_full; continue
        if light_min>0:
            md=done&miss
            if md.sum()>=light_min:
                n=md.sum(); rays+=n
                rem[md]=ray_len(n); miss[md]=rng.random(n)<p_miss
                cost+=C_light; continue
        rem[act]-=1; act_sum+=act.sum(); steps+=1; cost+=1.0
    return act_sum/(64*steps), cost/rays
base=sim()
print("base occupancy %.3f cost/ray %.4f"%base)
for lm in [8,12,16,24]:
    for cl in [1.5,2.5,3.5]:
        o,c=sim(light_min=lm,C_light=cl)
        print(f"light_min {lm:2d} C_light {cl}: occupancy {o:.3f} cost/ray {c:.4f} ({(c/base[1]-1)*100:+.1f}%)")
