// ntt.hpp -- NTT plan (cached twiddle tables per (log_n, omega)) and the pass driver.
#pragma once
#include <vector>

#include "common.hpp"

namespace h2 {

struct NttPlan {
    uint32_t log_n = 0;
    Fr w;                              // the root of unity the plan was built for (Montgomery form)
    std::vector<uint32_t> bits;        // B_p per pass
    Fr* tables = nullptr;              // one allocation: lo | hi | per-pass butterfly tables
    const Fr* tw_lo = nullptr;         // w^i,        i < min(n, 4096)
    const Fr* tw_hi = nullptr;         // w^(i<<12),  i < n >> 12
    std::vector<const Fr*> tw_bfly;    // per pass: (w^(n/R))^e, e < R/2
    std::vector<const Fr*> tw_direct;  // per pass: full inter-pass twiddle table or nullptr
    std::mutex mu;                     // guards scaled_hi, last_direct
    std::map<std::string, Fr*> scaled_hi;  // divisor -> tw_hi * divisor (iNTT: 1/n folded into the last pass)
    // the last pass's complete inter-pass twiddle set, w^(rho * K) at [(K << B_last) | rho] (2^log_n entries, streamed in
    // the order the pass loads its elements), keyed by the divisor folded into it ("" = none); nullptr = allocation failed
    std::map<std::string, Fr*> last_direct;
};

void ntt_split(uint32_t log_n, std::vector<uint32_t>& bits);
NttPlan* ntt_get_plan(DeviceCtx* ctx, uint32_t log_n, const uint64_t omega[4], hipStream_t stream);
// src (in_len valid elements, zero-extended to 2^log_n) -> dst; tmp = 2^log_n scratch (needed when
// the plan has >= 2 passes).  pre3 / post3: nullable HOST pointers to 3 Fr each (passed by value
// in the kernel arguments): x[i] *= pre3[i % 3] (i % 3 != 0) on load, y[i] *= post3[i % 3] on store.
void ntt_run(DeviceCtx* ctx, NttPlan* pl, const Fr* src, Fr* dst, Fr* tmp, uint32_t in_len, const Fr* pre3,
             const Fr* post3, hipStream_t stream);
Fr fr_from_u64x4(const uint64_t v[4]);

}  // namespace h2
