// search_noalloc.hip -- mz_search for callers that cannot allocate: the scratch is a file-scope device array.
#include <math.h>
#include <mutex>
#include "common.h"
#include "search_internal.h"

// ------------------------------------------------------------------------------------------------
// The same search for callers that cannot allocate (code under hipGraph capture after a first warm-up call, tight serving
// loops, harnesses that forbid hipMalloc): the workspace is a file-scope device array.  Every GPU the library runs on
// holds its own copy of the array, at its own address, until the process ends; host code never passes the array's NAME
// to a kernel or a memset -- it asks the runtime for the address once per device (hipGetSymbolAddress) and uses that.
//   scratch = [pb_c table, MZ_SEARCH_NOALLOC_MAX_SIMS + 2 entries, at a fixed place so it survives calls of other sizes]
//             [the workspace of mz_search for (n_trees, n_sims)]
// ------------------------------------------------------------------------------------------------
static __device__ __align__(256) unsigned char scratch[MZ_SEARCH_NOALLOC_BYTES];

namespace {
struct PbcEntry { double a, s; };
constexpr int kTabEntries = MZ_SEARCH_NOALLOC_MAX_SIMS + 2;
constexpr int64_t kTabBytes = ((int64_t)kTabEntries * 16 + 255) / 256 * 256;
constexpr int kMaxDevices = 64;
struct StaticDev {
    char *ptr;            // device address of THIS device's copy of `scratch`
    bool have_table;
    double base, init;    // the pb_c knobs the resident table was computed for
};
StaticDev g_dev[kMaxDevices];
PbcEntry g_tab[kTabEntries];   // host staging of the table (static: no allocation); guarded by g_mu
std::mutex g_mu;
}  // namespace

extern "C" int mz_search_noalloc(const mz_tower_weights *repr, const mz_tower_weights *dyn, const mz_heads_weights *heads,
                                 const mz_search_cfg *cfg, const void *image, const int8_t *to_play, const uint64_t *legal,
                                 const double *noise, int32_t n_sims, int32_t n_trees, int32_t *visits, float *root_value,
                                 void *stream) {
    if (int rc = mz_check_search_args("mz_search_noalloc", repr, dyn, heads, cfg, image, to_play, legal, n_sims, n_trees, visits, root_value)) return rc;
    if (n_trees == 0) return MZ_OK;
    MZ_REQUIRE(n_sims <= MZ_SEARCH_NOALLOC_MAX_SIMS, "mz_search_noalloc: n_sims %d exceeds the static limit %d", n_sims,
               MZ_SEARCH_NOALLOC_MAX_SIMS);
    MZ_REQUIRE(cfg->pb_c_base > 0, "mz_search_noalloc: pb_c_base must be > 0");
    SearchPlan P;
    if (int rc = mz_plan_search(n_trees, n_sims, kTabBytes, 0, &P)) return rc;
    MZ_REQUIRE(P.total <= (int64_t)MZ_SEARCH_NOALLOC_BYTES,
               "mz_search_noalloc: %d trees x %d simulations need %lld bytes of scratch, the static array holds %lld "
               "(use mz_search with a caller-owned workspace)", n_trees, n_sims, (long long)P.total, (long long)MZ_SEARCH_NOALLOC_BYTES);
    int dev = -1;
    MZ_HIP(hipGetDevice(&dev));
    MZ_REQUIRE(dev >= 0 && dev < kMaxDevices, "mz_search_noalloc: device ordinal %d not supported", dev);
    hipStream_t s = (hipStream_t)stream;
    char *ws;
    {
        std::lock_guard<std::mutex> lock(g_mu);
        StaticDev &d = g_dev[dev];
        if (!d.ptr) {   // once per device, with that device current
            void *p = nullptr;
            MZ_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
            MZ_REQUIRE(p && ((uintptr_t)p & 255) == 0, "mz_search_noalloc: bad scratch address");
            d.ptr = (char *)p;
        }
        ws = d.ptr;
        if (!d.have_table || d.base != cfg->pb_c_base || d.init != cfg->pb_c_init) {
            // ucb_score main.py:154-156 with the host's libm, as mz_pool_init does; only on a device's first call and
            // when the pb_c knobs change (one small synchronous upload -- make that call outside any graph capture)
            for (int n = 0; n < kTabEntries; ++n) {
                g_tab[n].a = log(((double)n + cfg->pb_c_base + 1) / cfg->pb_c_base) + cfg->pb_c_init;
                g_tab[n].s = sqrt((double)n);
            }
            MZ_HIP(hipMemcpyAsync(ws, g_tab, sizeof(g_tab), hipMemcpyHostToDevice, s));
            MZ_HIP(hipStreamSynchronize(s));   // g_tab is reused by the next caller
            d.have_table = true;
            d.base = cfg->pb_c_base;
            d.init = cfg->pb_c_init;
        }
    }
    // the array keeps the last call's contents.  Nothing in it is read before this call has written it (root_init writes
    // every tree's meta record, root records and root hidden state; a node's records, hidden state and path entries are
    // written by the simulation that creates them), but the meta records are cleared anyway, as mz_pool_init does.
    MZ_HIP(hipMemsetAsync(ws + P.L.off_meta, 0, (size_t)n_trees * 64, s));
    return mz_run_search(repr, dyn, heads, cfg, image, to_play, legal, noise, n_sims, n_trees, ws, P, visits, root_value, stream);
}
