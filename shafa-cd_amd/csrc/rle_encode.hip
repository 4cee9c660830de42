#include "common.hpp"
#include "internal.hpp"
int rleenc_launch(Batch *, hipStream_t, int, const u8 *, const u64 *, const u64 *, u8 *, const u64 *, const u64 *, u64 *, u64 *) { return SHAFA_OUTSIDE_MODULE; }
