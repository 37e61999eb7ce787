// kernels of the rosen likelihood (see hens_ktable.h)
#define HENS_KT_LIKE LIKE_ROSEN
#define HENS_KT_NAME rosen
#include "hens_ktable.inc"
