/* PTAR: a tiny named-array container used by the test infrastructure.
 *
 * Layout: magic "PTAR1\n", then records
 *   u32 name_len | name bytes | u32 dtype | u64 count | raw little-endian data
 * dtype: 0=f64 1=i32 2=f32 3=u8 4=i64
 *
 * Header-only C; shared by the reference driver, the oracle tools and the
 * product's dump tool so that arrays can be compared bit for bit.
 */
#ifndef PTAR_H
#define PTAR_H
#include <stdint.h>
#include <stdio.h>
#include <string.h>

enum { PTAR_F64 = 0, PTAR_I32 = 1, PTAR_F32 = 2, PTAR_U8 = 3, PTAR_I64 = 4 };

static inline size_t ptar_dtype_size(uint32_t dt) {
  switch (dt) {
    case PTAR_F64: return 8;
    case PTAR_I32: return 4;
    case PTAR_F32: return 4;
    case PTAR_U8: return 1;
    case PTAR_I64: return 8;
  }
  return 0;
}

static inline FILE *ptar_open_write(const char *path) {
  FILE *f = fopen(path, "wb");
  if (f) fwrite("PTAR1\n", 1, 6, f);
  return f;
}

static inline void ptar_put(FILE *f, const char *name, uint32_t dtype,
                            const void *data, uint64_t count) {
  uint32_t nl = (uint32_t)strlen(name);
  fwrite(&nl, 4, 1, f);
  fwrite(name, 1, nl, f);
  fwrite(&dtype, 4, 1, f);
  fwrite(&count, 8, 1, f);
  if (count) fwrite(data, ptar_dtype_size(dtype), count, f);
}

#endif
