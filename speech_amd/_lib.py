"""ctypes binding of libspeech.so (the C ABI of include/codee_speech.h and include/speech_hip.h).

The library is loaded from speech_amd/lib/ (built in-tree by speech_amd/csrc/Makefile).  There is no
fallback of any kind: if the library is missing, loading raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SPEECH_AMD_LIB") or os.path.join(_HERE, "lib", "libspeech.so")   # the override is for A/B timing of builds

KINDS = {"spectrogram": 0, "fbank": 1, "mfcc": 2, "plp": 3}
F32, F64 = 0, 1

LEGACY_SYMBOLS = [
    "kaldi_fbank_s", "kaldi_fbank_d", "kaldi_fbank_default_s", "kaldi_fbank_default_d",
    "kaldi_mfcc_s", "kaldi_mfcc_d", "kaldi_plp_s", "kaldi_plp_d",
    "kaldi_spectrogram_s", "kaldi_spectrogram_d",
]
EXT_SYMBOLS = [
    "speech_opts_default", "speech_device_count", "speech_last_error",
    "speech_plan_create", "speech_plan_destroy", "speech_plan_sizes", "speech_plan_num_frames",
    "speech_batch_create", "speech_batch_destroy", "speech_batch_totals", "speech_batch_offsets",
    "speech_batch_run", "speech_batch_run_timed", "speech_batch_run_pcm16",
    "speech_batch_stats", "speech_batch_cmvn", "speech_batch_affine", "speech_batch_add_deltas", "speech_batch_splice",
    "speech_batch_splice_subsample", "speech_batch_subsample_rows", "speech_extract_pcm16", "speech_batch_set_ids", "speech_plan_run_host", "speech_host_release",
    "speech_batch_cmvn_noalloc", "speech_run_host_multi", "speech_shard_utterances", "speech_opts_frames", "speech_set_dropin_dither", "speech_batch_synth_audio", "speech_synth_audio_host",
]


class SpeechOpts(C.Structure):
    """Mirror of `struct speech_opts` (include/speech_hip.h)."""
    _fields_ = [
        ("kind", C.c_int), ("dtype", C.c_int),
        ("sr", C.c_double), ("frame_length", C.c_double), ("frame_shift", C.c_double),
        ("snip_edges", C.c_int), ("dither", C.c_double), ("dc0", C.c_int), ("raw_energy", C.c_int),
        ("preemph", C.c_double), ("win_type", C.c_char * 32),
        ("amp", C.c_int), ("lof", C.c_double), ("hif", C.c_double), ("B", C.c_size_t), ("lg", C.c_int),
        ("use_energy", C.c_int), ("C", C.c_size_t), ("Q", C.c_double),
        ("compress", C.c_double), ("cep_scale", C.c_double), ("lpc_order", C.c_size_t),
        ("mn0", C.c_int), ("dither_seed", C.c_ulonglong), ("force_generic", C.c_int),
    ]


_lib = None


def load():
    """Load libspeech.so and declare every prototype.  Raises OSError if the library is missing."""
    global _lib
    if _lib is not None:
        return _lib
    # One process must hold ONE HIP runtime.  PyTorch-ROCm ships its own libamdhip64.so.7; if torch is importable,
    # load it first so that libspeech.so (DT_NEEDED libamdhip64.so.7) binds to the same copy instead of /opt/rocm's.
    # Two initialised copies fail with "No HIP GPUs are available" in whichever comes second.
    if not os.environ.get("SPEECH_AMD_NO_TORCH"):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(LIB_PATH):
        raise OSError("libspeech.so not found at %s: build it with `make -C speech_amd/csrc` "
                      "(or __graft_entry__.build()); there is no fallback path" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    i, z, s, vp = C.c_int, C.c_size_t, C.c_char_p, C.c_void_p
    pz = C.POINTER(C.c_size_t)
    for sfx, r in (("s", C.c_float), ("d", C.c_double)):
        p = C.POINTER(r)
        front = [p, p, z, r, r, r, i, r, i, i, r, s]
        getattr(lib, "kaldi_spectrogram_" + sfx).argtypes = front + [i]
        getattr(lib, "kaldi_fbank_" + sfx).argtypes = front + [i, r, r, z, i, i, i]
        getattr(lib, "kaldi_fbank_default_" + sfx).argtypes = [p, p, z, r, i, r, i, i, r, s, i, r, r, i, i, i]
        getattr(lib, "kaldi_mfcc_" + sfx).argtypes = front + [r, r, z, z, r, i, i]
        getattr(lib, "kaldi_plp_" + sfx).argtypes = front + [r, r, z, r, z, r, r, i, i]
    for name in LEGACY_SYMBOLS:
        getattr(lib, name).restype = i
    po = C.POINTER(SpeechOpts)
    lib.speech_opts_default.argtypes = [po, i]; lib.speech_opts_default.restype = None
    lib.speech_device_count.argtypes = []; lib.speech_device_count.restype = i
    lib.speech_last_error.argtypes = []; lib.speech_last_error.restype = s
    lib.speech_plan_create.argtypes = [C.POINTER(vp), po, i]; lib.speech_plan_create.restype = i
    lib.speech_plan_destroy.argtypes = [vp]; lib.speech_plan_destroy.restype = None
    lib.speech_plan_sizes.argtypes = [vp, pz, pz, pz, pz, C.POINTER(i)]; lib.speech_plan_sizes.restype = i
    lib.speech_plan_num_frames.argtypes = [vp, z, pz]; lib.speech_plan_num_frames.restype = i
    lib.speech_batch_create.argtypes = [C.POINTER(vp), vp, z, pz, pz, pz]; lib.speech_batch_create.restype = i
    lib.speech_batch_destroy.argtypes = [vp]; lib.speech_batch_destroy.restype = None
    lib.speech_batch_totals.argtypes = [vp, pz, pz, pz]; lib.speech_batch_totals.restype = i
    lib.speech_batch_offsets.argtypes = [vp, pz, pz, pz]; lib.speech_batch_offsets.restype = i
    lib.speech_batch_run.argtypes = [vp, vp, vp, vp]; lib.speech_batch_run.restype = i
    lib.speech_batch_run_timed.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_float)]; lib.speech_batch_run_timed.restype = i
    lib.speech_batch_run_pcm16.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_float)]; lib.speech_batch_run_pcm16.restype = i
    pd = C.POINTER(C.c_double)
    lib.speech_batch_stats.argtypes = [vp, vp, pd, pd, C.POINTER(C.c_ulonglong), vp]; lib.speech_batch_stats.restype = i
    lib.speech_batch_cmvn.argtypes = [vp, vp, i, vp]; lib.speech_batch_cmvn.restype = i
    lib.speech_batch_affine.argtypes = [vp, vp, pd, pd, vp]; lib.speech_batch_affine.restype = i
    lib.speech_batch_cmvn_noalloc.argtypes = [vp, vp, i, vp]; lib.speech_batch_cmvn_noalloc.restype = i
    lib.speech_synth_audio_host.argtypes = [C.c_ulonglong, C.c_ulonglong, z, C.POINTER(C.c_short)]; lib.speech_synth_audio_host.restype = i
    lib.speech_set_dropin_dither.argtypes = [C.c_ulonglong, C.c_longlong]; lib.speech_set_dropin_dither.restype = None
    lib.speech_batch_synth_audio.argtypes = [vp, vp, i, C.c_ulonglong, C.POINTER(C.c_ulonglong), vp]; lib.speech_batch_synth_audio.restype = i
    lib.speech_batch_add_deltas.argtypes = [vp, vp, vp, i, i, vp]; lib.speech_batch_add_deltas.restype = i
    lib.speech_batch_splice.argtypes = [vp, vp, vp, i, i, vp]; lib.speech_batch_splice.restype = i
    lib.speech_batch_splice_subsample.argtypes = [vp, vp, vp, i, i, i, i, vp]; lib.speech_batch_splice_subsample.restype = i
    lib.speech_batch_subsample_rows.argtypes = [vp, i, i, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]; lib.speech_batch_subsample_rows.restype = i
    lib.speech_batch_set_ids.argtypes = [vp, C.POINTER(C.c_uint)]; lib.speech_batch_set_ids.restype = i
    lib.speech_plan_run_host.argtypes = [vp, z, C.POINTER(vp), C.POINTER(z), C.POINTER(vp), i, z, C.POINTER(C.c_uint)]; lib.speech_plan_run_host.restype = i
    lib.speech_host_release.argtypes = []; lib.speech_host_release.restype = None
    lib.speech_run_host_multi.argtypes = [po, i, C.POINTER(i), z, C.POINTER(vp), C.POINTER(z), C.POINTER(vp), i, z, C.POINTER(C.c_uint)]; lib.speech_run_host_multi.restype = i
    lib.speech_shard_utterances.argtypes = [z, C.POINTER(z), i, C.POINTER(i)]; lib.speech_shard_utterances.restype = i
    lib.speech_opts_frames.argtypes = [po, z, pz, pz]; lib.speech_opts_frames.restype = i
    lib.speech_extract_pcm16.argtypes = [C.POINTER(SpeechOpts), C.POINTER(C.c_short), z, vp]; lib.speech_extract_pcm16.restype = i
    _lib = lib
    return lib


def last_error():
    return (load().speech_last_error() or b"").decode()
