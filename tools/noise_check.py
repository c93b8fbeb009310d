import sys, os
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import signals, oracle_lib as O, np_truth as T
from compare import rms_err, max_err
from speech_amd import api
for o in (dict(sr=22050.0, fl=16.352270085665104, fs=10.0, snip=0, dc0=0, raw_energy=1, preemph=0.97, win="rectangular"),
          dict(sr=16000.0, fl=19.054831460212846, fs=12.5, snip=1, dc0=0, raw_energy=1, preemph=0.0, win="rectangular"),
          dict(sr=16000.0, fl=25.0, fs=10.0, snip=1, dc0=1, raw_energy=1, preemph=0.97, win="povey")):
    x = signals.utterance(6605, 5000 + 3, sr=o["sr"], dtype=np.float32)
    truth = T.spectrogram(np.asarray(x, dtype=np.float64), single=True, **o)
    _, ref = O.run("spectrogram", x, single=True, **o)
    _, ref_fast = O.run("spectrogram", x, single=True, exact=False, **o)
    res = {"oracle exact-fft": ref, "oracle float-fft": ref_fast}
    for fg in (0, 1):
        plan = api.Plan("spectrogram", force_generic=fg, **o)
        b = api.Batch(plan, [x.size]); Y = torch.zeros(b.total_out, device="cuda"); b.run(torch.from_numpy(x).cuda(), Y); torch.cuda.synchronize()
        res["hip tuned" if plan.tuned else "hip generic"] = Y.cpu().numpy().reshape(-1, plan.D)
    print(o)
    for k, v in res.items():
        print("   %-18s rms %.3e max %.3e" % (k, rms_err("spectrogram", o, v, truth), max_err("spectrogram", o, v, truth)))
