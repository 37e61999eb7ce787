// kernels of the dense likelihood (see hens_ktable.h)
#define HENS_KT_LIKE LIKE_DENSE
#define HENS_KT_NAME dense
#include "hens_ktable.inc"
