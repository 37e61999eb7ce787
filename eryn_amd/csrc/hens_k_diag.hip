// kernels of the diag likelihood (see hens_ktable.h)
#define HENS_KT_LIKE LIKE_DIAG
#define HENS_KT_NAME diag
#include "hens_ktable.inc"
