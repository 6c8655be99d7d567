/*
 * ser_bit.c -- oracle (test infrastructure only): wire primitives of the reference.
 * Restates src/ser.rs (little-endian ints, length-prefixed slices) and src/bit.rs
 * (MSB-first IoBitWriter with zero padding).
 */
#include "cniic_oracle.h"
#include <stdlib.h>
#include <string.h>

void orc_buf_init(orc_buf *b) { b->data = NULL; b->len = 0; b->cap = 0; }
void orc_buf_free(orc_buf *b) { free(b->data); b->data = NULL; b->len = b->cap = 0; }

int orc_buf_put(orc_buf *b, const void *p, size_t n) {
    if (b->len + n > b->cap) {
        size_t nc = b->cap ? b->cap * 2 : 256;
        while (nc < b->len + n) nc *= 2;
        uint8_t *nd = (uint8_t *)realloc(b->data, nc);
        if (!nd) return ORC_ERR_NOMEM;
        b->data = nd;
        b->cap = nc;
    }
    memcpy(b->data + b->len, p, n);
    b->len += n;
    return ORC_OK;
}

/* ser.rs:17-21 */
int orc_ser_u8(orc_buf *b, uint8_t v) { return orc_buf_put(b, &v, 1); }
/* ser.rs:31-35: to_le_bytes */
int orc_ser_u16(orc_buf *b, uint16_t v) {
    uint8_t t[2] = {(uint8_t)v, (uint8_t)(v >> 8)};
    return orc_buf_put(b, t, 2);
}
/* ser.rs:49-53 */
int orc_ser_i16(orc_buf *b, int16_t v) { return orc_ser_u16(b, (uint16_t)v); }
/* ser.rs:67-71 */
int orc_ser_u32(orc_buf *b, uint32_t v) {
    uint8_t t[4];
    for (int i = 0; i < 4; i++) t[i] = (uint8_t)(v >> (8 * i));
    return orc_buf_put(b, t, 4);
}
/* ser.rs:87-91; usize is normalised to u64 (ser.rs:113-117) */
int orc_ser_u64(orc_buf *b, uint64_t v) {
    uint8_t t[8];
    for (int i = 0; i < 8; i++) t[i] = (uint8_t)(v >> (8 * i));
    return orc_buf_put(b, t, 8);
}
/* ser.rs:210-214: Rgb<T> serialises as the slice &self.0[..] => ser.rs:164-172:
 * u64 length (3) followed by the elements. 8 + 3 = 11 bytes. */
int orc_ser_rgb(orc_buf *b, const uint8_t rgb[3]) {
    int rc = orc_ser_u64(b, 3);
    if (rc) return rc;
    return orc_buf_put(b, rgb, 3);
}

int orc_de_u8(orc_rd *r, uint8_t *v) {
    if (r->pos >= r->n) return ORC_ERR_DECODE;
    *v = r->p[r->pos++];
    return ORC_OK;
}
int orc_de_u16(orc_rd *r, uint16_t *v) {
    if (r->pos + 2 > r->n) { r->pos = r->n; return ORC_ERR_DECODE; }
    *v = (uint16_t)(r->p[r->pos] | (r->p[r->pos + 1] << 8));
    r->pos += 2;
    return ORC_OK;
}
int orc_de_i16(orc_rd *r, int16_t *v) {
    uint16_t u;
    int rc = orc_de_u16(r, &u);
    *v = (int16_t)u;
    return rc;
}
int orc_de_u32(orc_rd *r, uint32_t *v) {
    if (r->pos + 4 > r->n) { r->pos = r->n; return ORC_ERR_DECODE; }
    uint32_t x = 0;
    for (int i = 0; i < 4; i++) x |= (uint32_t)r->p[r->pos + i] << (8 * i);
    r->pos += 4;
    *v = x;
    return ORC_OK;
}
int orc_de_u64(orc_rd *r, uint64_t *v) {
    if (r->pos + 8 > r->n) { r->pos = r->n; return ORC_ERR_DECODE; }
    uint64_t x = 0;
    for (int i = 0; i < 8; i++) x |= (uint64_t)r->p[r->pos + i] << (8 * i);
    r->pos += 8;
    *v = x;
    return ORC_OK;
}
/* ser.rs:216-222: Vec<T> then try_into [T;3] (fails unless len == 3) */
int orc_de_rgb(orc_rd *r, uint8_t rgb[3]) {
    uint64_t len;
    if (orc_de_u64(r, &len)) return ORC_ERR_DECODE;
    if (len != 3) return ORC_ERR_DECODE;
    for (int i = 0; i < 3; i++)
        if (orc_de_u8(r, &rgb[i])) return ORC_ERR_DECODE;
    return ORC_OK;
}

/* ---------------- bit.rs ---------------- */

/* bit.rs:103-105: ((1u16 << nbits) - 1) as u8 */
uint8_t orc_bit_mask(uint8_t nbits) { return (uint8_t)(((uint32_t)1 << nbits) - 1); }

/* bit.rs:70-86 */
int orc_bit_nth(uint8_t byte, uint8_t idx, int msb_first) {
    uint8_t mask = msb_first ? (uint8_t)(0x80 >> idx) : (uint8_t)(1u << idx);
    return (byte & mask) ? 1 : 0;
}

void orc_bitw_init(orc_bitw *w, orc_buf *out) {
    w->out = out;
    w->curr_bits = 0;
    w->bit_count = 0;
}

/* bit.rs:210-220 (push_bit MsbFirst, bit.rs:46-48) */
int orc_bitw_bit(orc_bitw *w, int bit) {
    w->curr_bits = (uint8_t)((w->curr_bits << 1) | (bit & 1));
    w->bit_count++;
    if (w->bit_count == 8) {
        int rc = orc_buf_put(w->out, &w->curr_bits, 1);
        if (rc) return rc;
        w->bit_count = 0;
    }
    return ORC_OK;
}

/* bit.rs:222-240 */
int orc_bitw_byte(orc_bitw *w, uint8_t n) {
    if (w->bit_count == 0) return orc_buf_put(w->out, &n, 1);
    uint8_t msb = (uint8_t)(w->curr_bits << (8 - w->bit_count));
    uint8_t lsb = (uint8_t)(n >> w->bit_count);
    uint8_t done = msb | lsb;
    int rc = orc_buf_put(w->out, &done, 1);
    if (rc) return rc;
    w->curr_bits = n & orc_bit_mask(w->bit_count);
    return ORC_OK;
}

/* bit.rs:164-178 write_arr over a BitArray{full_bytes, partial_byte, partial_count} */
int orc_bitw_code(orc_bitw *w, const uint8_t *full_bytes, size_t nfull, uint8_t partial_byte,
                  uint8_t partial_count) {
    for (size_t i = 0; i < nfull; i++) {
        int rc = orc_bitw_byte(w, full_bytes[i]);
        if (rc) return rc;
    }
    for (uint8_t j = (uint8_t)(8 - partial_count); j < 8; j++) {
        int rc = orc_bitw_bit(w, orc_bit_nth(partial_byte, j, 1));
        if (rc) return rc;
    }
    return ORC_OK;
}

/* bit.rs:243-253 */
int orc_bitw_pad_and_flush(orc_bitw *w) {
    if (w->bit_count != 0) {
        w->curr_bits = (uint8_t)(w->curr_bits << (8 - w->bit_count));
        int rc = orc_buf_put(w->out, &w->curr_bits, 1);
        if (rc) return rc;
        w->curr_bits = 0;
        w->bit_count = 0;
    }
    return ORC_OK;
}
