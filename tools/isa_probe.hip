// Tuning aid: the forward step (fwd_issue + fwd_finish of csrc/lane_steps.hpp) as a kernel of its own, to count its instructions.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DPA_PROBE_FAST_ONLY] -Irust-pseudoaligner_amd/csrc --cuda-device-only -S tools/isa_probe.hip -o /tmp/probe.s
//   python tools/isa_count.py /tmp/probe.s k_fwd          (-DPA_PROBE_FAST_ONLY: the straight-line step without the general one)
#include <hip/hip_runtime.h>
#include "lane_steps.hpp"
using namespace pa;
__global__ void k_fwd(Lane* st, const DevIndexView ix, const uint64_t* rd, uint32_t* cols, uint32_t allowed) {
    Lane s = st[threadIdx.x];
    const ReadRef rr{rd + threadIdx.x, 64, 5};
    uint32_t* c = cols + 64 * threadIdx.x;
    const ColRef cr{c, c + 4, c + 8, c + 12, c + 16, c + 20, 32, c + 20, nullptr};
    FwdLoad f;
    fwd_issue(s, ix, f);
    fwd_finish<false>(s, ix, rr, cr, allowed, f);
    st[threadIdx.x] = s;
}
