"""What orc_params::pair_manifold (Bullet's persistent <= 4-point manifold for link-link / link-box pairs, oracle only) changes:
rollouts of the reference's scenarios with the switch off and on, the joint-3 reaction by gait phase, and random folded snakes.
    python tools/pair_manifold_effect.py     (CPU only; the numbers of DESIGN.md 3 / profiles/r04_pair_manifold.txt)"""
import sys, json, time
sys.path.insert(0,'oracle'); sys.path.insert(0,'.')
import numpy as np, oracle as orc, bench
GAIT_TEST_WORLD = dict(dt=0.01, gravity_z=-9.81, max_motor_impulse=4.0 * 0.01)
def rollout(over, J=60, phi=0.0, scale=1.0):
    e = orc.OracleEnv(**over); e.reset()
    peak=0.0; subs=0; rew=0.0; dones=0; ncmax=0; pairpts=0; pairsteps=0
    b0 = e.get_box()[0].copy() if over.get('obstacle') else None
    for j in range(J):
        k=np.arange(8)
        a=-np.sin((2*k+1)*4.0+2.0*(0.1*j)+phi)*scale
        o,r,d,kk,_ = e.env_step(a, vec_mode=True)
        subs+=kk; rew+=r; dones+=int(d)
        peak=max(peak, abs(e.joint3_reaction_fz()))
        c=e.last_contacts_full()
        if len(c):
            npair=int((c[:,5]!=-1).sum())  # linkB: -1 ground, -2 box? see dump
            pairpts+=npair; pairsteps+=1 if npair else 0
    b1 = e.get_box()[0] if over.get('obstacle') else None
    disp = float(np.linalg.norm(b1[:3]-b0[:3])) if b0 is not None else 0.0
    return dict(substeps=subs/J, reward=rew/J, dones=dones, x=float(e.get_state()[0]), peak_f3=peak, box_disp_mm=1e3*disp, pair_points_last_substeps=pairpts, steps_with_pair_points=pairsteps)
t=time.time()
for name, over, kw in [
    ("training world, gait (no obstacle)", dict(), dict(J=40)),
    ("training world, free box 0.1 m ahead", dict(obstacle=2, obstacle_pos=[0.1,0.0,0.1]), dict(J=60)),
    ("gait-test world, free box 0.1 m ahead", dict(obstacle=2, obstacle_pos=[0.1,0.0,0.1], **GAIT_TEST_WORLD), dict(J=60)),
    ("training world, static box 0.1 m ahead", dict(obstacle=1, obstacle_pos=[0.1,0.0,0.1]), dict(J=60)),
]:
    for pm in (0,1):
        r = rollout(dict(pair_manifold=pm, **over), **kw)
        print(name, "| pair_manifold", pm, "|", json.dumps({k:(round(v,5) if isinstance(v,float) else v) for k,v in r.items()}))
print("seconds", time.time()-t)
print("--- gait-test world, peaks and means of |joint-3 reaction| over the env-steps where the box is touched, by phase")
for phi in (0.0, 0.7, 1.4, 2.1, 2.8, 3.5):
    out=[]
    for pm in (0,1):
        e = orc.OracleEnv(pair_manifold=pm, obstacle=2, obstacle_pos=[0.1,0.0,0.1], **GAIT_TEST_WORLD); e.reset()
        f=[]
        for j in range(60):
            k=np.arange(8); a=-np.sin((2*k+1)*4.0+2.0*(0.1*j)+phi)
            e.env_step(a, vec_mode=True)
            c=e.last_contacts_full()
            if len(c) and (c[:,5]!=-1).any(): f.append(abs(e.joint3_reaction_fz()))
        f=np.array(f) if f else np.zeros(1)
        out.append((len(f), float(f.max()), float(f.mean()), float(np.percentile(f,90))))
    print("phi %.1f | off: n %d peak %.1f mean %.2f p90 %.1f | on: n %d peak %.1f mean %.2f p90 %.1f" % ((phi,)+out[0]+out[1]))

# ---- random folded snakes (|q| <= 1.7): how often link-link contacts occur at all, and how far the two schemes drift apart
sys.path.insert(0,'tests')
import numpy as np, oracle as orc
from conftest import random_state
rng=np.random.default_rng(77)
n=16; B=200
res={5:[],20:[]}; npts=[]
e0=orc.OracleEnv(pair_manifold=0); e1=orc.OracleEnv(pair_manifold=1)
for i in range(B):
    s=random_state(rng,n,z=0.026,qamp=1.7,vamp=0.3,flat=True); s[9]*=0.1; s[7:9]*=0.1
    t=rng.uniform(-0.5,0.5,n)
    for e in (e0,e1):
        e.hard_reset(); e.set_state(s)
    for k in range(1,21):
        e0.substep(t); e1.substep(t)
        if k in res:
            a=e0.get_state(); b=e1.get_state()
            res[k].append((np.abs(a[13+n:]-b[13+n:])/(1+np.abs(a[13+n:]))).max())
    c0=e0.last_contacts_full(); c1=e1.last_contacts_full()
    npts.append(((c0[:,5]>=0).sum() if len(c0) else 0, (c1[:,5]>=0).sum() if len(c1) else 0))
npts=np.array(npts)
for k in res:
    r=np.array(res[k]); print("K=%d: states differing %d of %d | rel velocity difference median %.2e p90 %.2e max %.2e" % (k,(r>0).sum(),B,np.median(r),np.percentile(r,90),r.max()))
print("link-link contact points at the last substep: stateless mean %.2f, manifold mean %.2f; states with any %d / %d" % (npts[:,0].mean(), npts[:,1].mean(), (npts[:,0]>0).sum(), (npts[:,1]>0).sum()))
