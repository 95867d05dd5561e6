/* slane_interp.c -- TEST INFRASTRUCTURE ONLY (see oracle.h): a CPU interpreter of the scan-lane sweep's programs
 * (omr-img-corrector_amd/csrc/slane.hpp).  It executes one strip's program exactly as a wave does -- the 65-register
 * ring, fetches committed one row late, (mask, pk) segments over register pairs -- for `lanes` scans held in the
 * interleaved bit image, and returns the strip's contribution to both projections.  tests/ compare that with the
 * oracle's own sweep (oracle.c, projection.rs:47-65): it pins the PROGRAM FORMAT and the generator on the CPU,
 * before any GPU is involved.  Nothing here is shipped or measured. */
#include <stdint.h>
#include <string.h>

#define SL_FETCH 4 /* landing registers per row: two pairs */
#define SL_FREC 4
#define SL_AHEAD 4
#define SL_TURN 16
#define SL_RING 68 /* 64 ring registers, register 64 holds 0 (65 is its upper neighbour), the pair 66 / 67 swallows dummy fetch pairs */

/* seg: n_records rows of seg_dwords dwords (K words x S segments, pk = sh | idx << 5 | 0x60000 | q' << 21 | n << 26 in the
 * first; format v4: the first segment names the register pair (idx - 1, idx) and is read straight into the word, q' is the
 * funnel shift of the NEXT slot -- slane.hpp); fet: n_records rows of 4 dwords (format v4): 2 byte offsets (E << 8) of the first entries of the PAIRS whose loads are issued in this row
 * -- a pair = entries e and e + 1, two adjacent word columns of one source row, into an aligned pair of landing registers --,
 * 1 dword = 2 x u16 (EVEN ring register | 0x8000): where the two pairs that landed in this row's set of landing
 * registers (the loads of SL_AHEAD rows earlier) are committed before this row, as the kernel's v_mov_b64 does (an odd
 * index would be rounded down by the hardware: refused here), 1 dword = the turn header: the slots per word the kernel
 * executes during the SL_TURN rows from a turn's first record on (the slots past a word's own count are the generator's
 * pads: they must be no-ops, and a header smaller than a word's count would drop segments); bits: entries[E][lanes] dwords
 * (entry 0 all zero);
 * hrow[r * lanes + lane] += black pixels of row r in this strip (r = record - pre_rows, 0 <= r < rows);
 * vcol[(k * 32 + bit) * lanes + lane] += black pixels of bit `bit` of the strip's word k (the caller maps bits to columns).
 * Returns 0, or -1 when a fetch / register index is out of range or a virtual row carries bits. */
int orc_slane_run_strip(const uint32_t *seg, int seg_dwords, const uint32_t *fet, int n_records, int pre_rows, int rows,
                        int K, const uint32_t *bits, int64_t n_entries, int lanes, uint32_t *hrow, uint32_t *vcol)
{
    const int S = seg_dwords / K;
    uint32_t ring[SL_RING][64], T[SL_AHEAD][SL_FETCH][64];
    int turn_slots = 0;
    if (lanes > 64) return -1;
    memset(ring, 0, sizeof ring);
    memset(T, 0, sizeof T);
    for (int q = 0; q < n_records; q++) {
        const uint32_t *rec = fet + (int64_t)q * SL_FREC;
        const int set = q % SL_AHEAD;
        /* commit what landed in this row's set of landing registers, then issue this row's loads into it */
        for (int pr = 0; pr < SL_FETCH / 2; pr++) {
            const uint32_t m0 = (rec[2] >> (16 * pr)) & 0xffffu; /* M0 image: register | DST_REL */
            const uint32_t reg = m0 & 255u;
            if (reg + 1 >= SL_RING || (reg & 1u) || reg == 64u || (m0 >> 8) != 0x80u) return -1;
            memcpy(ring[reg], T[set][2 * pr], sizeof(uint32_t) * (size_t)lanes);
            memcpy(ring[reg + 1], T[set][2 * pr + 1], sizeof(uint32_t) * (size_t)lanes);
        }
        if (q % SL_TURN == 0) { /* the kernel reads the header at the first record of a turn */
            turn_slots = (int)rec[3];
            if (turn_slots < 1 || turn_slots > S) return -1;
        }
        for (int pr = 0; pr < SL_FETCH / 2; pr++) {
            const int64_t e = rec[pr] >> 8;
            if (e >= n_entries || (rec[pr] & 255u)) return -1;
            memcpy(T[set][2 * pr], bits + e * lanes, sizeof(uint32_t) * (size_t)lanes);
            if (e + 1 < n_entries) memcpy(T[set][2 * pr + 1], bits + (e + 1) * lanes, sizeof(uint32_t) * (size_t)lanes);
            else memset(T[set][2 * pr + 1], 0, sizeof(uint32_t) * (size_t)lanes);
        }
        for (int k = 0; k < K; k++) {
            const uint32_t *w = seg + (int64_t)q * seg_dwords + k * S;
            const int n = (int)((w[0] >> 26) & 31u);
            if (n < 1 || n > S || n > turn_slots) return -1; /* the kernel would stop before this word's last segment */
            for (int lane = 0; lane < lanes; lane++) {
                uint32_t D = 0;
                for (int j = 0; j < turn_slots; j++) { /* exactly the header's slots, pads included */
                    const uint32_t pk = w[j];
                    const uint32_t idx = (pk >> 5) & 255u, sh = pk & 31u;
                    if (((pk >> 5) & 0xff00u) != 0x3000u || idx + 1 >= SL_RING) return -1; /* M0 image: index | SRC0_REL | SRC1_REL */
                    if (j == 0) { /* the first segment: the pair (index - 1, index) read straight into the word; register -1 is a
                                     landing register -- whatever it holds must not matter */
                        const uint32_t lo = idx ? ring[idx - 1][lane] : 0xdeadbeefu ^ (uint32_t)(q * 2654435761u + (uint32_t)lane);
                        D = (uint32_t)((((uint64_t)ring[idx][lane] << 32) | lo) >> sh);
                    } else { /* the funnel shift's amount rides in the slot before */
                        const uint32_t qq = (w[j - 1] >> 21) & 31u;
                        const uint64_t pair = ((uint64_t)ring[idx + 1][lane] << 32) | ring[idx][lane];
                        const uint32_t X = (uint32_t)(pair >> sh);
                        D = (uint32_t)((((uint64_t)X << 32) | D) >> qq);
                    }
                }
                if (q >= pre_rows && q - pre_rows < rows) hrow[(int64_t)(q - pre_rows) * lanes + lane] += (uint32_t)__builtin_popcount(D);
                else if (D) return -1; /* virtual rows carry no bits */
                for (int b = 0; b < 32; b++) vcol[(int64_t)(k * 32 + b) * lanes + lane] += (D >> b) & 1u;
            }
        }
    }
    return 0;
}
