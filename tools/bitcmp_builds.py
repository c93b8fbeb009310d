"""Bit-compare two builds of libspeech.so on generic-path workloads (one library per child process):
python tools/bitcmp_builds.py libA.so libB.so"""
import os, sys, subprocess, numpy as np
code = '''
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import signals
from speech_amd import api
out = []
for kind, kw, dt in (("mfcc", dict(sr=8000.0, hif=4000.0, frame_length=10.0), np.float32), ("spectrogram", {}, np.float64), ("fbank", dict(sr=22050.0, hif=11025.0, B=40), np.float32),
                     ("plp", {}, np.float64), ("mfcc", dict(sr=44100.0, hif=20000.0), np.float64), ("mfcc", dict(mn0=1, raw_energy=0, use_energy=1), np.float64)):
    x = signals.utterance(30000, 5, sr=kw.get("sr", 16000.0)).astype(dt)
    out.append(getattr(api, "kaldi_" + kind)(x, **kw))
np.savez(sys.argv[1], *out)
'''
for i, lib in enumerate(sys.argv[1:3]):
    env = dict(os.environ, SPEECH_AMD_LIB=os.path.abspath(lib))
    subprocess.check_call([sys.executable, "-c", code, "/tmp/bitcmp_%d.npz" % i], env=env)
a, b = np.load("/tmp/bitcmp_0.npz"), np.load("/tmp/bitcmp_1.npz")
for k in a.files:
    print(k, a[k].shape, "bit-identical" if np.array_equal(a[k], b[k]) else "DIFFERENT max %.3g" % np.abs(a[k] - b[k]).max())
