"""SHA-256 over the sources of the tuned (profiled) kernels: fast_*.hip, fast_common.hpp, speech_internal.hpp and the Makefile.  tools/pmc_to_json.py stores it beside the
counters of a profile; bench.py recomputes it and reports the counters only while the sources they were measured on are the
sources of the library it runs (otherwise the counter fields are null and `counters_stale` is true)."""
import glob
import hashlib
import os

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")


def kernel_sources_sha():
    h = hashlib.sha256()
    for path in sorted(glob.glob(os.path.join(CSRC, "fast_*.hip")) + [os.path.join(CSRC, n) for n in ("fast_common.hpp", "speech_internal.hpp", "Makefile")]):
        h.update(os.path.basename(path).encode() + b"\0")
        h.update(open(path, "rb").read())
    return h.hexdigest()[:16]
