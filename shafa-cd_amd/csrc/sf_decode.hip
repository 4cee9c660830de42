#include "common.hpp"
#include "internal.hpp"
int sfdec_launch(Batch *, hipStream_t, int, const u8 *, const u64 *, const u64 *, const shafa_code_table *, const u64 *, u8 *, const u64 *) { return SHAFA_OUTSIDE_MODULE; }
