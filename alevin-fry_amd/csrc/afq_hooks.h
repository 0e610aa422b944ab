// afq_hooks.h — the ONE place the library reads test hooks from the environment.
//
// AFQ_TEST_<NAME>: routing overrides and sizes that tests/ flips to reach every code path on inputs of a few thousand reads
// (which decoder, which parsimony route, how small a slab or a pool, where a range is cut).  A hook changes WHICH path
// computes the rows, never the rows: every one of them is exercised by a parity test against the oracle.  Rounds 1-4 had
// grown 36 getenv() sites, two thirds of them measurement switches whose alternative had been measured and not kept; those
// alternatives are gone (a measurement build is `make variant DEFS=...`), and what is left besides the hooks is:
//   AFQ_EM_ORDER=canonical   the sequential f32 EM of rounds 1-3, bit-identical to the reference's arithmetic
//   AFQ_HOST_TIMING=1        where the host side of a batch spends its time, on stderr
#pragma once
#include <stdlib.h>
#include <string.h>

namespace afq {

inline const char* test_hook(const char* name) {
    char k[64] = "AFQ_TEST_";
    strncat(k, name, sizeof(k) - strlen(k) - 1);
    return getenv(k);
}
inline long test_hook_long(const char* name, long dflt) { const char* e = test_hook(name); return e ? atol(e) : dflt; }
inline bool test_hook_is(const char* name, const char* v) { const char* e = test_hook(name); return e && !strcmp(e, v); }
inline bool em_order_canonical() { const char* e = getenv("AFQ_EM_ORDER"); return e && !strcmp(e, "canonical"); }

}  // namespace afq
