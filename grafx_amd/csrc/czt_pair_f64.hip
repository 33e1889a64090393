// The double-precision instances of czt_pair.hip (gfx_odd_alias_pair_precise_*) as a translation unit of their own: the
// column kernels are instantiated per tile count (35 sizes) and precision, and one file with both took four minutes to build.
#define GFX_CZT_PAIR_F64
#include "czt_pair.hip"
