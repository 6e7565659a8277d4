#include "common.h"
#include "../../include/maggie_hip.h"
extern "C" int mg_abi_version(void) { return 1; }

#include <utility>
#include <vector>

char* mg_zeroed_lo = nullptr;
char* mg_zeroed_hi = nullptr;
static std::vector<std::pair<char*, char*>> g_claims;      // the slices of the range that an accumulator has already been "cleared" in
static int g_conflicts = 0;
/* [base, base + bytes) is zero and every part of it is handed to at most one accumulator before it is zeroed again (NULL / 0: none).
 * mg_zero_words() inside the library then skips its fill launch for buffers inside the range. Single-threaded use (the capturing thread). */
extern "C" int mg_set_zeroed_range(void* base, long bytes) {
    mg_zeroed_lo = (char*)base;
    mg_zeroed_hi = base ? (char*)base + bytes : nullptr;
    g_claims.clear();
    return 0;
}
/* The invariant behind the skipped fills, checked: a second "clear" of words that an earlier call already claimed inside the current range means
 * that an accumulator is being re-zeroed after use (or two callees share a slice) -- under capture the fill would be skipped and stale sums would
 * survive. The skip is then refused (the caller's entry point fails with hipErrorAlreadyMapped) instead of trusted. */
extern "C" int mg_zero_claim(void* p, long bytes) {
    char* lo = (char*)p;
    char* hi = lo + bytes;
    for (const auto& c : g_claims)
        if (lo < c.second && c.first < hi) { ++g_conflicts; return 1; }
    g_claims.emplace_back(lo, hi);
    return 0;
}
extern "C" int mg_zeroed_range_conflicts(void) { const int n = g_conflicts; g_conflicts = 0; return n; }
