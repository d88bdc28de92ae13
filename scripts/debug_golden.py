import sys; sys.path.insert(0,'.')
import numpy as np, torch
from safemotionsrisk_amd import VecSafeMotionsEnv, SPACE_ENV_CONFIG
from oracle.oracle import OracleEnv
g=np.load('tests/golden/space_step.npz'); B=int(g['B']); T=int(g['T'])
env=VecSafeMotionsEnv(B, seed=11, auto_reset=False, **SPACE_ENV_CONFIG)
env.set_state(q=g["q0"], v=g["v0"], a=g["a0"], planet_idx=g["planet_idx0"], step_count=g["step_count0"], moving_hit=g["moving_hit0"])
ora=OracleEnv(env.scene,1)
for t in range(T):
    st0=env.get_state()
    obs, rew, done, info = env.step(torch.from_numpy(g["actions"][t]).cuda())
    obs=obs.cpu().numpy()
    err=np.abs(obs-g['obs'][t])
    e,c=np.unravel_index(np.argmax(err),err.shape)
    print('t',t,'max err pos/vel/acc/planet',err[:,:7].max(),err[:,7:14].max(),err[:,14:21].max(),err[:,21:].max(),'worst env',e,'col',c,'gpu',obs[e,c],'gold',g['obs'][t][e,c],'action',g['actions'][t][e,c%7])
    if err.max()>3e-4:
        j=c%7
        l,h,cd=ora.safe_acc_range(st0['q'][e].astype(np.float64),st0['v'][e].astype(np.float64),st0['a'][e].astype(np.float64))
        gl,gh,gc=env.test_safe_acc_range(st0['q'][e:e+1],st0['v'][e:e+1],st0['a'][e:e+1])
        print('  state p,v,a',st0['q'][e,j],st0['v'][e,j],st0['a'][e,j],'oracle range',l[j],h[j],cd[j],'gpu range',gl[0,j],gh[0,j],gc[0,j], 'lim', env.scene.struct.pos_hi[j], env.scene.struct.vel_max[j])
    gs=g['state'][t]; st=env.get_state()
    env.set_state(q=gs[:, :7].astype(np.float32), v=gs[:, 7:14].astype(np.float32), a=gs[:, 14:].astype(np.float32),
                  step_count=np.where(g["done"][t], 0, st["step_count"]), moving_hit=np.where(g["done"][t], 0, st["moving_hit"]))
