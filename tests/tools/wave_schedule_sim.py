"""Monte-Carlo of one wave's search / service schedule (not a pytest; CPU only): rays of Gamma(2) length with mean 13 visits, a third of
them ending in a miss; the wave leaves the search loop when fewer than `thr` lanes search (full service pass, cost C_full steps)
or - the "light" service pass of DESIGN.md 9 - when `light_min` lanes wait with a ray that left the scene (cost C_light).
Prints search-loop occupancy and cost per ray. Calibrated to tests/tools/step_probe.py on C3 (0.60 occupied, service a quarter of
the time). Result: the light pass never pays, whatever it costs between 1.5 and 3.5 steps.
Round 5 (`python wave_schedule_sim.py paths`): a second path per lane - LANE-PRIVATE (sim_private: a lane whose search is over swaps to its own
parked ray; the review's candidate ii) and WAVE-WIDE (sim_pool: any idle lane takes any ready ray of a 64-slot pool; what csrc/sol_pool.hip
builds). In this model (a turn costs 1 whoever takes part, a shading pass 4.4, an exchange pass 0.3) the private form is worth -2 .. +2 %, the
pool -7 .. -10 % of the cost per ray; on the GPU the pool kernel is 7 % SLOWER (profiles/r05_pool_kernel_ab.txt): a turn with more lanes in it
costs more (its primitive part runs more often), the exchange passes cost ~150 instructions each, the shading passes get thinner."""
import sys
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


def sim_private(thr=16, K=8, G=1.5, S=4.4, C_swap=0.3, steps_total=40000):
    """Two contexts per lane, the second parked lane-privately (1 = ready ray, 2 = finished hit)."""
    rem=ray_len(64); parked=np.ones(64,int); act_sum=0; steps=0; cost=0.0; rays=0
    while steps<steps_total:
        act=rem>0; can=(~act)&(parked==1); ns=can.sum()
        if ns>=K or (ns>0 and act.sum()<thr):
            rem[can]=ray_len(ns); parked[can]=2; cost+=C_swap; continue
        if act.sum()<thr:
            done=~act; both=done&(parked==2); n=done.sum(); rays+=n+both.sum()
            rem[done]=ray_len(n); parked[both]=1
            cost+=G+S+((S+C_swap) if both.any() else 0.0); continue
        rem[act]-=1; act_sum+=act.sum(); steps+=1; cost+=1.0
    return act_sum/(64*steps), cost/rays
def sim_pool(P=64, thr=16, K=8, G=1.5, S=4.4, C_swap=0.3, steps_total=40000):
    """A wave-wide pool of P parked contexts: an idle lane exchanges its finished search for ANY ready ray; the service block shades what the
    lanes hold, then exchanges fresh rays for parked hits pass by pass."""
    rem=ray_len(64); n_ray=P; n_hit=0; act_sum=0; steps=0; cost=0.0; rays=0
    while steps<steps_total:
        act=rem>0; fin=~act; F=fin.sum()
        if n_ray>0 and (F>=K or (F>0 and act.sum()<thr)):
            m=min(F,n_ray); idx=np.flatnonzero(fin)[:m]; rem[idx]=ray_len(m); n_ray-=m; n_hit+=m; cost+=C_swap; continue
        if act.sum()<thr:
            c=G+S; rays+=F
            while n_hit>0:
                m=min(n_hit,F); n_hit-=m; n_ray+=m; rays+=m; c+=S+C_swap
            cost+=c; rem[fin]=ray_len(F); continue
        rem[act]-=1; act_sum+=act.sum(); steps+=1; cost+=1.0
    return act_sum/(64*steps), cost/rays
if len(sys.argv)>1 and sys.argv[1]=="paths":
    print("two paths per lane (cost per ray against base %.4f, occupancy %.3f):"%(base[1],base[0]))
    for thr in (16,24,32):
        for K in (8,16):
            o,c=sim_private(thr=thr,K=K); o2,c2=sim_pool(thr=thr,K=K)
            print(f"  thr {thr:2d} K {K:2d}: lane-private occupancy {o:.3f} cost {(c/base[1]-1)*100:+.1f}% | wave-wide pool occupancy {o2:.3f} cost {(c2/base[1]-1)*100:+.1f}%")
