// Non-kernel entry points of the C ABI (include/bevrender_hip.h).
#include "bevr_common.h"

extern "C" int bevr_abi_version(void) { return BEVR_ABI_VERSION; }

extern "C" const char* bevr_strerror(int code) {
  switch (code) {
    case BEVR_OK: return "ok";
    case BEVR_E_NULL: return "a required pointer is NULL";
    case BEVR_E_SHAPE: return "a dimension violates the documented contract";
    case BEVR_E_PRECISION: return "unknown precision code";
    case BEVR_E_ALIGN: return "a pointer is not sufficiently aligned";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown bevrender error";
  }
}

extern "C" int bevr_attn_table_dims(bevr_attn_desc* d) {
  if (!d) return BEVR_E_NULL;
  if (d->S < 2 || d->Wt < 1) return BEVR_E_SHAPE;
  d->Sp = 32 * ((d->S + 31) / 32);
  d->Ht = 2 * d->S - 1;
  d->y_off = d->Sp + 2;
  d->Hp = d->Ht + 2 * d->Sp + 4;
  const int half = d->Wt / 2;  // ceil((Wt - 1) / 2)
  d->x_off = half + 4;
  d->Wp = d->Wt + 2 * half + 9;
  return BEVR_OK;
}
