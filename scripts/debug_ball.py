import sys, time; sys.path.insert(0,'.')
import numpy as np, torch
from safemotionsrisk_amd import VecSafeMotionsEnv, BALL_ENV_CONFIG
from oracle.oracle import OracleEnv
KEYS=VecSafeMotionsEnv.STATE_KEYS
B,T=4096,30
env=VecSafeMotionsEnv(B, seed=5, auto_reset=False, **BALL_ENV_CONFIG)
t=time.time(); obs0=env.reset().cpu().numpy(); print('reset s',time.time()-t, obs0.shape)
ora=OracleEnv(env.scene,B,seed=5,num_threads=16)
oobs=ora.reset(); ost=ora.get_state(); st=env.get_state()
same=(np.abs(ost['q']-st['q']).max(1)<1e-3)&(ost['ball_i']==st['ball_i']).all(1)
print('reset agreement',same.mean(),'ball diff',np.abs(ost['ball']-st['ball'])[same].max(),'obs diff',np.abs(oobs-obs0)[same].max())
bad=np.nonzero(~same)[0][:3]
for e in bad: print(' env',e,'gpu ball_i',st['ball_i'][e],'ora',ost['ball_i'][e],'q diff',np.abs(ost['q'][e]-st['q'][e]).max(),'ball',st['ball'][e],ost['ball'][e])
rng=np.random.default_rng(9); reasons={}
for t in range(T):
    st=env.get_state(); ora.set_state(**{k: st[k] for k in KEYS})
    act=rng.uniform(-1,1,(B,7)).astype(np.float32)
    obs,rew,done,info=env.step(torch.from_numpy(act).cuda())
    o_obs,o_rew,o_done,o_reason,o_info=ora.step(act.astype(np.float64))
    obs=obs.cpu().numpy(); rew=rew.cpu().numpy(); vals=info['values'].cpu().numpy(); done=done.cpu().numpy(); reason=info['termination_reason'].cpu().numpy()
    st1=env.get_state(); ost1=ora.get_state()
    good=(done==o_done)&(reason==o_reason)
    bi=(st1['ball_i']==ost1['ball_i']).all(1)
    err=np.abs(obs-o_obs)*(good&bi)[:,None]
    for x in reason[done>0]: reasons[int(x)]=reasons.get(int(x),0)+1
    print('t',t,'obs err kin',err[:,:21].max(),'ball',err[:,21:].max(),'rew err',(np.abs(rew-o_rew)*good).max(),'flag mismatch',(~good).sum(),'ball_i mismatch',(~bi).sum(),'spawned',(st1['ball_i'][:,4]!=st['ball_i'][:,4]).sum(), 'd_moving err', (np.abs(vals[:,2]-o_info[:,2])*good).max())
    for e in np.nonzero(~bi)[0][:2]: print('   env',e,'gpu',st1['ball_i'][e],'ora',ost1['ball_i'][e],'pre',st['ball_i'][e])
    if done.any(): env.reset(mask=torch.from_numpy(done).cuda())
print(reasons)
for Bb in (65536,):
    e2 = VecSafeMotionsEnv(Bb, seed=0, auto_reset=True, return_info=False, **BALL_ENV_CONFIG)
    e2.reset(); act = torch.rand((Bb,7), device='cuda')*2-1
    for _ in range(5): e2.step(act)
    e2.synchronize(); e2.set_profiling(True); t=time.time(); n=50
    for _ in range(n): e2.step(act)
    e2.synchronize(); dt=time.time()-t
    pr=e2.get_profile()
    print(Bb, 'ball env-steps/s', Bb*n/dt, {k:round(v['ms']/max(v['launches'],1),3) for k,v in pr.items()})
