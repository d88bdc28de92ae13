import sys; sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, torch
from oracle.oracle import OracleEnv
import test_gpu_parity as T
B, NT = 4096, 16
env = T._ball_env(B, seed=5)
env.reset()
ora = OracleEnv(env.scene, B, seed=5, num_threads=16)
ora.reset()
keys = env.STATE_KEYS
rng = np.random.default_rng(9)
for t in range(NT):
    st = env.get_state()
    ora.set_state(**{k: st[k] for k in keys})
    act = rng.uniform(-1, 1, (B, 7)).astype(np.float32)
    obs, rew, done, info = env.step(torch.from_numpy(act).cuda())
    o_obs, o_rew, o_done, o_reason, o_info = ora.step(act.astype(np.float64))
    rew = rew.cpu().numpy(); done = done.cpu().numpy()
    reason = info["termination_reason"].cpu().numpy()
    st1, ost1 = env.get_state(), ora.get_state()
    good = (done == o_done) & (reason == o_reason) & (st1["ball_i"] == ost1["ball_i"]).all(axis=1)
    err = np.abs(rew - o_rew); err[~good] = 0
    i = int(err.argmax())
    if err[i] > 2e-5:
        names = [k for k in info.keys()]
        print('step', t, 'env', i, 'err', err[i], 'rew', rew[i], o_rew[i])
        gi = env.info[i].cpu().numpy()
        print(' gpu info', gi); print(' ora info', o_info[i])
        print(' q', st['q'][i], 'v', st['v'][i], 'a', st['a'][i])
        np.savez('gpurun_out/ball_case.npz', q=st1['q'][i] if False else st['q'][i], v=st['v'][i], a=st['a'][i], a1=st1['a'][i], ball=st['ball'][i], ball_i=st['ball_i'][i], ginfo=gi, oinfo=o_info[i])
        print(' next a gpu', st1['a'][i], 'ora', ost1['a'][i], 'act', act[i])
    if done.any():
        env.reset(mask=torch.from_numpy(done).cuda())
