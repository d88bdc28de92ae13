import sys; sys.path.insert(0,'.')
import numpy as np, torch
from safemotionsrisk_amd import VecSafeMotionsEnv, SPACE_ENV_CONFIG
from oracle.oracle import OracleEnv
KEYS=("q", "v", "a", "planet_idx", "step_count", "moving_hit", "rng_draws")
B,T=4096,12
env=VecSafeMotionsEnv(B, seed=5, auto_reset=False, **SPACE_ENV_CONFIG); env.reset()
ora=OracleEnv(env.scene,B,seed=5,num_threads=16)
rng=np.random.default_rng(9)
for t in range(T):
    st=env.get_state(); ora.set_state(**{k: st[k] for k in KEYS})
    act=rng.uniform(-1,1,(B,7)).astype(np.float32)
    obs,rew,done,info=env.step(torch.from_numpy(act).cuda())
    o_obs,o_rew,o_done,o_reason,o_info=ora.step(act.astype(np.float64))
    rew=rew.cpu().numpy(); vals=info['values'].cpu().numpy(); done=done.cpu().numpy()
    good=(done==o_done)
    err=np.abs(rew-o_rew)*good
    e=int(np.argmax(err))
    print('t',t,'max rew err',err.max(),'env',e,'gpu info',vals[e,:3],'ora info',o_info[e,:3], 'mismatch', (~good).sum())
    if err.max()>2e-3:
        st1=env.get_state()
        q=st1['q'][e].astype(np.float64); idx=st1['planet_idx'][e]
        print('  q',st1['q'][e].tolist(),'idx',idx.tolist())
        print('  oracle distances at gpu state', ora.distances(q, idx), 'gpu test_distances', env.test_distances(st1['q'][e:e+1], idx[None]))
        best=None
        for link in range(3,8):
            for p in range(2):
                for c in range(24 if p==0 else 1):
                    d=ora.link_planet_piece_distance(q,link,p,c,int(idx[p]))
                    if best is None or d<best[0]: best=(d,link,p,c)
        print('  closest pair (oracle)',best)
    if done.any(): env.reset(mask=torch.from_numpy(done).cuda())
