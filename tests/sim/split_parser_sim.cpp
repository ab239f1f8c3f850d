// Host build of the split decoder's PARSER (lz4_flex_amd/csrc/lz4_split_parser.h, -DLZ4FLEX_HOST_SIM): one lane
// walks a block's token chain exactly as on the device, its records are executed by a plain byte-wise copier,
// and tests/test_split_parser_sim.py compares bytes, length and error variant with the oracle.  Test
// infrastructure only: nothing in the product loads this.
#include <stdint.h>
#include <string.h>

#include <vector>

#include "../../lz4_flex_amd/csrc/lz4_split_parser.h"

using namespace lz4flex_dev::v5;

// returns the status code (0 ok); *out_len = bytes produced; detail[0..1] = expected, actual for OutputTooSmall.
// pad_before: the block is copied to an address with this misalignment (exercises the aligned-space window);
// every byte outside [0, in_len) of the private copy is poisoned and never legally read.
template <class L>
static int sim_run(const uint8_t* in, uint32_t in_len, uint8_t* out, uint32_t cap, uint32_t* out_len,
                   uint64_t* detail, uint32_t misalign, uint32_t* n_records, uint32_t* n_steps) {
    constexpr uint32_t BLK_LDS = L::BLK_LDS, TAIL_OFF = L::TAIL_OFF;
    std::vector<uint8_t> lds(BLK_LDS, 0);
    // private copy: [poison 64][block][poison 64]; the parser may read up to 3 bytes before an unaligned block
    std::vector<uint8_t> buf(64 + 4 + in_len + 64, 0xEE);
    uint8_t* gin = buf.data() + 64;
    gin += (4 - ((uintptr_t)gin & 3)) & 3;
    gin += misalign & 3;
    if (in_len) memcpy(gin, in, in_len);
    ParserT<L> p;
    p.q.blk = lds.data();
    p.q.set_head(0);
    p.q.set_tail(0);
    p.init_window(gin, in_len);
    p.rare_below = (misalign & 4u) ? 8u : 4u;
    p.lit_slack = (misalign & 4u) ? 15u : 3u;
    p.cap = cap;
    for (uint32_t i = 0; i < TAIL_BUF; ++i) lds[TAIL_OFF + i] = (p.tstart + i < in_len) ? gin[p.tstart + i] : 0;
    p.ip = 0; p.op = 0; p.tok_over = 0; p.qtail = 0; p.status = 0; p.expected = 0; p.done = 0;
    lz4_sim_chaos_state = (misalign & 16u) ? 0x9E3779B9u ^ in_len ^ (cap << 7) : 0u;
    p.prime();
    if (in_len == 0) p.fail(LZ4FLEX_DEV_E_EXPECTED_ANOTHER_BYTE);
    uint32_t head = 0, op = 0, recs = 0, steps = 0;
    bool finished = false;
    for (;;) {
        // drain the queue (the copier side, byte-wise)
        while (head != p.q.tail()) {
            const u32x4 e = p.q.get(head);
            head++;
            p.q.set_head(head);
            recs++;
            const uint32_t lsrc = e.x, ln = e.y, ml = e.z, off = e.w & 0xFFFFu;
            if (ln) {
                if ((uint64_t)lsrc + ln > in_len || (uint64_t)op + ln > cap) return -1000;   // protocol violation
                if ((e.w >> 16) < R_CAREFUL && (uint64_t)lsrc + ln + p.lit_slack > in_len) return -1001;   // wild reads must stay inside
                memcpy(out + op, gin + lsrc, ln);
                op += ln;
            }
            if (ml) {
                if (off == 0 || off > op || (uint64_t)op + ml > cap) return -1002;
                if ((off < p.rare_below) != ((e.w >> 16) == R_RARE)) return -1006;   // periodic matches must be marked
                for (uint32_t i = 0; i < ml; ++i) out[op + i] = out[op - off + i];
                op += ml;
            }
            if ((e.w >> 16) == R_FINISH) finished = true;
        }
        if (finished || p.done) {
            if (head == p.q.tail()) break;
            continue;
        }
        p.step();
        if (++steps > 40u * (in_len + 16u)) return -1003;   // no progress
    }
    if (!finished) return -1004;
    *out_len = p.status == 0 ? p.op : 0u;
    if (p.status == 0 && p.op != op) return -1005;
    detail[0] = p.status == LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL ? p.expected : 0;
    detail[1] = p.status == LZ4FLEX_DEV_E_OUTPUT_TOO_SMALL ? cap : 0;
    if (n_records) *n_records = recs;
    if (n_steps) *n_steps = steps;
    return p.status;
}

// misalign: bits 0-1 source misalignment, bit 2 a wide copier (offsets < 8 are rare, 16-byte words: 15 bytes of literal slack), bit 3 the small LDS layout (8-record queue), bit 4 the wave-level "any lane" tests fire at random (other lanes of the wavefront)
extern "C" int split_parser_sim(const uint8_t* in, uint32_t in_len, uint8_t* out, uint32_t cap, uint32_t* out_len,
                                uint64_t* detail, uint32_t misalign, uint32_t* n_records, uint32_t* n_steps) {
    return (misalign & 8u) ? sim_run<LayoutSmall>(in, in_len, out, cap, out_len, detail, misalign, n_records, n_steps)
                           : sim_run<LayoutBig>(in, in_len, out, cap, out_len, detail, misalign, n_records, n_steps);
}
