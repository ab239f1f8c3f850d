/*
 * lz4flex_block.c -- ORACLE (test infrastructure, never shipped): plain-C restatement of
 * lz4_flex's block encoder and decoder, 64-bit little-endian behaviour.
 *
 * Follows (paths relative to /root/reference):
 *   encoder  src/block/compress.rs:318-489 (compress_internal), :224-247, :252-287,
 *            :554-583, :588-590;  src/block/hashtable.rs:19-34,52-53,75-132
 *   decoder  src/block/decompress.rs:201-449 (unsafe flavour: the one the north star names;
 *            check order OffsetOutOfBounds before OutputTooSmall for matches),
 *            cross-read with src/block/decompress_safe.rs:93-318
 *   consts   src/block/mod.rs:35-77
 *
 * See lz4flex_oracle.h for the pinning status.  Not a copy: the reference is Rust; this is
 * a from-the-spec restatement in C written for this repository.
 */
#include "lz4flex_oracle.h"

#include <stdlib.h>
#include <string.h>

/* src/block/mod.rs:35-70 */
#define WINDOW_SIZE 65536u
#define MFLIMIT 12u
#define LAST_LITERALS 5u
#define END_OFFSET (LAST_LITERALS + 1u)
#define LZ4_MIN_LENGTH (MFLIMIT + 1u)
#define MAX_DISTANCE 65535u
#define MINMATCH 4u
/* src/block/compress.rs:28 */
#define INCREASE_STEPSIZE_BITSHIFT 5

static inline uint32_t rd32(const uint8_t *p) { uint32_t v; memcpy(&v, p, 4); return v; }
static inline uint64_t rd64(const uint8_t *p) { uint64_t v; memcpy(&v, p, 8); return v; }

/* src/block/hashtable.rs:19-21 */
static inline uint32_t hash4(uint32_t sequence) { return (sequence * 2654435761u) >> 16; }
/* src/block/hashtable.rs:27-34 (little-endian prime) */
static inline uint32_t hash5(uint64_t sequence) {
    return (uint32_t)(((sequence << 24) * 889523592379ull) >> 48);
}

/* The two live table kinds (hashtable.rs:58-92 and :96-132): 4096 entries, index = hash >> 4. */
typedef struct {
    int is_u16;        /* 1: HashTable4KU16 + hash4 ; 0: HashTable4K + hash5 */
    uint16_t t16[4096];
    uint32_t t32[4096];
} table_t;

static inline size_t tbl_hash_at(const table_t *t, const uint8_t *input, size_t pos) {
    /* hashtable.rs:88-91 (U16 override: 4-byte hash) / :40-44 (default: hash5 of 8 bytes) */
    return t->is_u16 ? hash4(rd32(input + pos)) : hash5(rd64(input + pos));
}
static inline size_t tbl_get(const table_t *t, size_t hash) {
    return t->is_u16 ? t->t16[hash >> 4] : t->t32[hash >> 4];
}
static inline void tbl_put(table_t *t, size_t hash, size_t val) {
    if (t->is_u16) t->t16[hash >> 4] = (uint16_t)val; else t->t32[hash >> 4] = (uint32_t)val;
}

/* src/block/compress.rs:588-590 */
size_t lz4o_get_maximum_output_size(size_t input_len) {
    return 16 + 4 + (size_t)((uint64_t)input_len * 110 / 100);
}

/* src/block/compress.rs:156-216.  The 8/4/2/1-byte ladder of the unsafe flavour computes the
 * common-prefix length bounded by input_end; restated byte-wise-equivalently with the same ladder. */
size_t lz4o_count_same_bytes(const uint8_t *input, size_t input_len, size_t *cur,
                             const uint8_t *source, size_t source_len, size_t candidate) {
    size_t max_input_match = (input_len >= *cur + END_OFFSET) ? input_len - (*cur + END_OFFSET) : 0;
    size_t max_candidate_match = source_len - candidate;
    size_t input_end = *cur + (max_input_match < max_candidate_match ? max_input_match : max_candidate_match);
    size_t start = *cur;
    const uint8_t *sp = source + candidate;
    while (*cur + 8 <= input_end) {
        uint64_t diff = rd64(input + *cur) ^ rd64(sp);
        if (diff == 0) { *cur += 8; sp += 8; }
        else { *cur += (size_t)(__builtin_ctzll(diff) / 8); return *cur - start; }
    }
    if (input_end - *cur >= 4) {
        uint32_t diff = rd32(input + *cur) ^ rd32(sp);
        if (diff == 0) { *cur += 4; sp += 4; }
        else { *cur += (size_t)(__builtin_ctz(diff) / 8); return *cur - start; }
    }
    if (input_end - *cur >= 2 && input[*cur] == sp[0] && input[*cur + 1] == sp[1]) { *cur += 2; sp += 2; }
    if (*cur < input_end && input[*cur] == sp[0]) { *cur += 1; }
    return *cur - start;
}

/* output cursor standing in for SliceSink (src/sink.rs:94-198) */
typedef struct { uint8_t *p; size_t pos, cap; } sink_t;
static inline void push_byte(sink_t *o, uint8_t b) { o->p[o->pos++] = b; }

/* src/block/compress.rs:224-233 */
static void write_integer(sink_t *o, size_t n) {
    while (n >= 0xFF) { n -= 0xFF; push_byte(o, 0xFF); }
    push_byte(o, (uint8_t)n);
}

/* src/block/compress.rs:237-247 */
static void handle_last_literals(sink_t *o, const uint8_t *input, size_t input_len, size_t start) {
    size_t lit_len = input_len - start;
    push_byte(o, lit_len < 0xF ? (uint8_t)(lit_len << 4) : 0xF0);           /* :65-74 */
    if (lit_len >= 0xF) write_integer(o, lit_len - 0xF);
    memcpy(o->p + o->pos, input + start, lit_len);
    o->pos += lit_len;
}

/* src/block/compress.rs:318-489 */
static int64_t compress_internal(const uint8_t *input, size_t input_len, size_t input_pos, sink_t *output,
                                 table_t *dict, int use_dict, const uint8_t *ext_dict, size_t ext_dict_len,
                                 size_t input_stream_offset) {
    /* :338-340 */
    if (output->cap - output->pos < lz4o_get_maximum_output_size(input_len - input_pos))
        return -LZ4O_E_OUTPUT_TOO_SMALL;
    size_t output_start_pos = output->pos;
    /* :343-346 */
    if (input_len - input_pos < LZ4_MIN_LENGTH) {
        handle_last_literals(output, input, input_len, input_pos);
        return (int64_t)(output->pos - output_start_pos);
    }
    size_t ext_dict_stream_offset = input_stream_offset - ext_dict_len;   /* :348 */
    size_t end_pos_check = input_len - MFLIMIT;                           /* :349 */
    size_t literal_start = input_pos;
    size_t cur = input_pos;
    /* :353-359 */
    if (cur == 0 && input_stream_offset == 0) {
        tbl_put(dict, tbl_hash_at(dict, input, 0), 0);
        cur = 1;
    }
    for (;;) {
        size_t step_size, candidate = 0, offset = 0;
        const uint8_t *candidate_source = input;
        size_t candidate_source_len = input_len;
        size_t non_match_count = (size_t)1 << INCREASE_STEPSIZE_BITSHIFT;
        size_t next_cur = cur;
        /* :373-439 probe loop */
        for (;;) {
            step_size = non_match_count >> INCREASE_STEPSIZE_BITSHIFT;
            non_match_count += 1;
            cur = next_cur;
            next_cur += step_size;
            if (cur > end_pos_check) {                                    /* :381-384 */
                handle_last_literals(output, input, input_len, literal_start);
                return (int64_t)(output->pos - output_start_pos);
            }
            size_t hash = tbl_hash_at(dict, input, cur);                  /* :391-393 */
            candidate = tbl_get(dict, hash);
            tbl_put(dict, hash, cur + input_stream_offset);
            if (input_stream_offset + cur - candidate > MAX_DISTANCE) continue;   /* :403-405 */
            if (candidate >= input_stream_offset) {                       /* :407-411 */
                offset = (uint16_t)(input_stream_offset + cur - candidate);
                candidate -= input_stream_offset;
                candidate_source = input; candidate_source_len = input_len;
            } else if (use_dict) {                                        /* :412-421 */
                offset = (uint16_t)(input_stream_offset + cur - candidate);
                candidate -= ext_dict_stream_offset;
                candidate_source = ext_dict; candidate_source_len = ext_dict_len;
            } else {                                                      /* :422-429 */
                continue;
            }
            if (rd32(candidate_source + candidate) == rd32(input + cur)) break;   /* :432-438 */
        }
        /* :442-448 backtrack_match (:272-287) */
        while (candidate > 0 && cur > literal_start && input[cur - 1] == candidate_source[candidate - 1]) {
            cur -= 1; candidate -= 1;
        }
        size_t lit_len = cur - literal_start;                             /* :451 */
        cur += MINMATCH; candidate += MINMATCH;                           /* :454-455 */
        size_t duplicate_length =
            lz4o_count_same_bytes(input, input_len, &cur, candidate_source, candidate_source_len, candidate);
        /* :460-461 */
        tbl_put(dict, tbl_hash_at(dict, input, cur - 2), cur - 2 + input_stream_offset);
        /* :463-486 emit */
        uint8_t token = (uint8_t)((lit_len < 0xF ? (lit_len << 4) : 0xF0) |
                                  (duplicate_length < 0xF ? duplicate_length : 0xF));   /* :77-96 */
        push_byte(output, token);
        if (lit_len >= 0xF) write_integer(output, lit_len - 0xF);
        memcpy(output->p + output->pos, input + literal_start, lit_len);  /* :478 (wild copy: same prefix) */
        output->pos += lit_len;
        push_byte(output, (uint8_t)(offset & 0xFF));                      /* :480 u16 LE */
        push_byte(output, (uint8_t)(offset >> 8));
        if (duplicate_length >= 0xF) write_integer(output, duplicate_length - 0xF);
        literal_start = cur;                                              /* :487 */
    }
}

/* src/block/compress.rs:571-583 */
static void init_dict(table_t *t, const uint8_t **dict_data, size_t *dict_len) {
    if (*dict_len > WINDOW_SIZE) { *dict_data += *dict_len - WINDOW_SIZE; *dict_len = WINDOW_SIZE; }
    size_t i = 0;
    while (i + 8 <= *dict_len) {
        tbl_put(t, tbl_hash_at(t, *dict_data, i), i);
        i += 3;
    }
}

/* src/block/compress.rs:554-568 */
static int64_t compress_into_sink_with_dict(int use_dict, const uint8_t *in, size_t in_len, sink_t *out,
                                            const uint8_t *dict_data, size_t dict_len) {
    table_t *t = (table_t *)calloc(1, sizeof(table_t));
    if (!t) return -LZ4O_E_OUTPUT_TOO_SMALL;
    t->is_u16 = (dict_len + in_len < 65535u);
    init_dict(t, &dict_data, &dict_len);
    int64_t r = compress_internal(in, in_len, 0, out, t, use_dict, dict_data, dict_len, dict_len);
    free(t);
    return r;
}

int64_t lz4o_compress_into(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap) {
    sink_t s = {out, 0, out_cap};
    return compress_into_sink_with_dict(0, in, in_len, &s, (const uint8_t *)"", 0);
}

int64_t lz4o_compress_into_with_dict(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                                     const uint8_t *dict, size_t dict_len) {
    sink_t s = {out, 0, out_cap};
    return compress_into_sink_with_dict(1, in, in_len, &s, dict, dict_len);
}

/* Independent-frame block, src/frame/compress.rs:280-299 + :357-367.
 * For k>0 the encoder's HashTable4K holds only entries < src_stream_offset, all of which
 * compress_internal skips (:403-405 or :422-429); emulate with stream offset = 1<<20 over a
 * zeroed table: every stale entry (0) is > MAX_DISTANCE away and the cur==0 seeding (:353)
 * does not fire, exactly as in the reference where src_stream_offset >= block_size > 0. */
int64_t lz4o_compress_frame_block(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                                  int first_block) {
    table_t *t = (table_t *)calloc(1, sizeof(table_t));
    if (!t) return -LZ4O_E_OUTPUT_TOO_SMALL;
    t->is_u16 = 0;
    sink_t s = {out, 0, out_cap};
    int64_t r = compress_internal(in, in_len, 0, &s, t, 0, (const uint8_t *)"", 0,
                                  first_block ? 0 : ((size_t)1 << 20));
    free(t);
    return r;
}

/* Exposed for the frame oracle (Linked mode needs the persistent table + prefix/ext-dict). */
void *lz4o__table_new(void) { table_t *t = (table_t *)calloc(1, sizeof(table_t)); if (t) t->is_u16 = 0; return t; }
void lz4o__table_free(void *t) { free(t); }
void lz4o__table_clear(void *tv) { table_t *t = (table_t *)tv; memset(t->t32, 0, sizeof t->t32); }
/* src/block/hashtable.rs:113-117 */
void lz4o__table_reposition(void *tv, uint32_t offset) {
    table_t *t = (table_t *)tv;
    for (int i = 0; i < 4096; i++) t->t32[i] = t->t32[i] > offset ? t->t32[i] - offset : 0;
}
int64_t lz4o__compress_internal(const uint8_t *input, size_t input_len, size_t input_pos, uint8_t *out,
                                size_t out_cap, void *table, int use_dict, const uint8_t *ext_dict,
                                size_t ext_dict_len, size_t input_stream_offset) {
    sink_t s = {out, 0, out_cap};
    return compress_internal(input, input_len, input_pos, &s, (table_t *)table, use_dict, ext_dict,
                             ext_dict_len, input_stream_offset);
}

/* ------------------------------------------------------------------------------------------ */
/* decoder */

/* src/block/decompress.rs:192-195 */
int lz4o_does_token_fit(uint8_t token) {
    return !(((token & 0x0F) == 0x0F) || ((token & 0xF0) == 0xF0));
}

/* src/block/decompress.rs:126-157 */
static int read_integer(const uint8_t *in, size_t in_len, size_t *ip, size_t *n_out) {
    size_t n = 0;
    for (;;) {
        if (*ip >= in_len) return -LZ4O_E_EXPECTED_ANOTHER_BYTE;
        uint8_t extra = in[(*ip)++];
        n += extra;
        if (extra != 0xFF) break;
    }
    *n_out = n;
    return 0;
}

/* src/block/decompress.rs:201-449.  The fast path (:259-328) is an optimisation of the same
 * semantics (it can only be entered when no bound can be violated), so the slow path (:330-444)
 * is restated for every token.  `out_pos` is the sink's initial pos (frame Linked prefix mode). */
int64_t lz4o__decompress_internal(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_pos, size_t out_cap,
                                  int use_dict, const uint8_t *ext_dict, size_t ext_dict_len,
                                  lz4o_err_detail *detail) {
    if (in_len == 0) return -LZ4O_E_EXPECTED_ANOTHER_BYTE;                 /* :207-209 */
    if (!use_dict) ext_dict_len = 0;
    size_t ip = 0, op = out_pos;
    for (;;) {
        uint8_t token = in[ip++];                                          /* :249-250 */
        size_t literal_length = token >> 4;                                /* :334 */
        if (literal_length != 0) {
            if (literal_length == 15) {                                    /* :336-340 */
                size_t ext; int e = read_integer(in, in_len, &ip, &ext);
                if (e) return e;
                literal_length += ext;
            }
            if (literal_length > in_len - ip) return -LZ4O_E_LITERAL_OUT_OF_BOUNDS;       /* :346-348 */
            if (literal_length > out_cap - op) {                                          /* :349-355 */
                if (detail) { detail->expected = op + literal_length; detail->actual = out_cap; }
                return -LZ4O_E_OUTPUT_TOO_SMALL;
            }
            memcpy(out + op, in + ip, literal_length);                     /* :357-361 */
            op += literal_length; ip += literal_length;
        }
        if (ip >= in_len) break;                                           /* :366-368 */
        if (in_len - ip < 2) return -LZ4O_E_EXPECTED_ANOTHER_BYTE;         /* :373-375 */
        size_t offset = (size_t)in[ip] | ((size_t)in[ip + 1] << 8);        /* :377, :161-174 */
        ip += 2;
        if (offset == 0) return -LZ4O_E_OFFSET_ZERO;
        size_t match_length = MINMATCH + (token & 0xF);                    /* :386-391 */
        if (match_length == MINMATCH + 15) {
            size_t ext; int e = read_integer(in, in_len, &ip, &ext);
            if (e) return e;
            match_length += ext;
        }
        size_t output_len = op;                                            /* :395 */
        if (offset > output_len + ext_dict_len) return -LZ4O_E_OFFSET_OUT_OF_BOUNDS;      /* :399-401 */
        if (match_length > out_cap - op) {                                                /* :402-407 */
            if (detail) { detail->expected = output_len + match_length; detail->actual = out_cap; }
            return -LZ4O_E_OUTPUT_TOO_SMALL;
        }
        if (use_dict && offset > output_len) {                             /* :410-426, copy_from_dict :85-109 */
            size_t dict_offset = ext_dict_len + output_len - offset;
            size_t dict_match_length = match_length < ext_dict_len - dict_offset ? match_length
                                                                                 : ext_dict_len - dict_offset;
            memcpy(out + op, ext_dict + dict_offset, dict_match_length);
            op += dict_match_length;
            if (dict_match_length == match_length) {
                if (ip >= in_len) return -LZ4O_E_EXPECTED_ANOTHER_BYTE;
                continue;
            }
            match_length -= dict_match_length;
        }
        /* :431-437 duplicate(): byte-serial forward copy semantics (:57-82).  Like the reference's hot loop
         * (:259-328, 16-byte wild copies when nothing can run out of bounds) the common case copies 8 bytes at a
         * time: legal when source and destination chunks cannot overlap (offset >= 8) and the up to 7 bytes
         * written behind the match stay inside the sink; every other case keeps the checked byte loop. */
        {
            const uint8_t *src = out + op - offset;
            uint8_t *dst = out + op;
            if (offset >= 8 && match_length + 8 <= out_cap - op) {
                for (size_t i = 0; i < match_length; i += 8) memcpy(dst + i, src + i, 8);
            } else {
                for (size_t i = 0; i < match_length; i++) dst[i] = src[i];
            }
            op += match_length;
        }
        if (ip >= in_len) return -LZ4O_E_EXPECTED_ANOTHER_BYTE;            /* :439-443 */
    }
    return (int64_t)(op - out_pos);                                        /* :445-448 */
}

int64_t lz4o_decompress_into(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                             lz4o_err_detail *detail) {
    return lz4o__decompress_internal(in, in_len, out, 0, out_cap, 0, (const uint8_t *)"", 0, detail);
}

int64_t lz4o_decompress_into_with_dict(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_cap,
                                       const uint8_t *dict, size_t dict_len, lz4o_err_detail *detail) {
    return lz4o__decompress_internal(in, in_len, out, 0, out_cap, 1, dict, dict_len, detail);
}
