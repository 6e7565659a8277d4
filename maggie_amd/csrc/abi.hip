#include "common.h"
#include "../../include/maggie_hip.h"
extern "C" int mg_abi_version(void) { return 1; }

char* mg_zeroed_lo = nullptr;
char* mg_zeroed_hi = nullptr;
/* [base, base + bytes) is zero and every part of it is handed to at most one accumulator before it is zeroed again (NULL / 0: none).
 * mg_zero_words() inside the library then skips its fill launch for buffers inside the range. Single-threaded use (the capturing thread). */
extern "C" int mg_set_zeroed_range(void* base, long bytes) {
    mg_zeroed_lo = (char*)base;
    mg_zeroed_hi = base ? (char*)base + bytes : nullptr;
    return 0;
}
