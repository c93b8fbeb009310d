// fast_kernels.hip -- tuned gfx950 fp32 kernels (placeholder: nothing tuned yet, every plan uses generic_kernels.hip).
#include "speech_internal.hpp"

namespace speech {

struct FastTables { int unused; };

bool fast_supported(const speech_opts &, size_t, size_t) { return false; }
int fast_tile_frames(const speech_opts &) { return GENERIC_TILE_FRAMES; }
FastTables *fast_tables_create(const speech_opts &, size_t, size_t, size_t, size_t, size_t, size_t, const std::vector<float> &,
                               const std::vector<float> &, const std::vector<int> &, const std::vector<int> &, const std::vector<float> &,
                               const std::vector<float> &, const std::vector<float> &, const std::vector<float> &, std::string *err)
{
    if (err) { *err = "no tuned kernel"; }
    return nullptr;
}
void fast_tables_destroy(FastTables *t) { delete t; }
size_t fast_scratch_elems(const FastTables *, size_t) { return 0; }
hipError_t launch_fast(const FastTables *, const FrameParams<float> &, float *, hipStream_t, hipEvent_t, hipEvent_t) { return hipErrorNotSupported; }

}  // namespace speech
