"""Generates omr-img-corrector_amd/csrc/slane_asm.inc: the scan-lane sweep's wave program as
gfx950 assembly text; ONE asm statement that dispatches on the strip's segment-slot class (2 / 4 / 8 slots per word laid out) and,
once per turn of 16 rows, on the turn's header (1 .. S slots executed: 2 + 4 + 8 loop bodies; program format v3, round 5).

Why assembly: the wave keeps its source words in a ring of 64 VGPRs addressed through the gfx9 VGPR index mode (M0),
its segment descriptors live in SGPRs filled by s_load, and the column counters are a carry-save tree in fixed
registers -- none of which HIP C++ can express (no dynamically indexed register arrays, no SGPR arrays).
tools/slane_mb.hip / slane_mb2.hip measured the pieces on the hardware.

A turn = four destination rows = two batches of two rows.  Scalar loads return out of order, so the only wait there is
is lgkmcnt(0) -- everything outstanding.  A record requested one row ahead therefore had exactly one row of flight time
(~100-200 ns against ~400 ns of scalar-cache miss): that wait was the largest single cost of the first version
(profiles/r04_lanes_ablation.md).  Hence four record sets and ONE wait per TWO rows: while rows 0, 1 are swept from
sets A, B the records of rows 2, 3 travel into C, D, and vice versa.  Program format v2 makes that fit the SGPR file:
a segment is ONE dword (slane.hpp) -- no masks, a word is assembled by funnel shifts
    X = v_alignbit(ring[idx + 1], ring[idx], sh)      (VGPR index mode: M0 = pk >> 5)
    D = X << q  (first segment)   |   D = v_alignbit(X, D, q)  (the others: X's low q bits enter at the top)
Per row:  wait until the loads issued four rows ago have landed (counted: vector loads return in order) -> commit them
into the ring (two v_mov_b64 with DST_REL: a row's two loads each bring a pair of adjacent word columns, tools/mov64_probe.hip
showed that a 64-bit move takes an even index) -> two words -> issue this row's loads: per pair ONE s_add_u32 moves the base of
the descriptor copy s[96:99] to the pair's first entry and two loads with the constant offsets 0 / 256 follow (format v4) -- a
vector-memory instruction whose offset comes out of an SGPR costs an issue turn more than one with a constant offset, 2.7 ms
per launch with four loads per row (profiles/r05_lanes_ablation.md); the images of the scan groups are placed so that the add
never carries (slane.hpp) -> row count (pairs of rows meet the
workgroup's other strips in LDS; one global atomic per pair row and block of BLOCK rows) -> (odd rows) carry-save column counters.

Register map (fixed; the statement clobbers s0-s13, s16-s101, v1-v127, so lane * 4 arrives in v0):
  s[0:1] segment stream  s[2:3] fetch stream  s[4:7] bit-image descriptor  s8 rows left (whole turns; the record NUMBERS end on a multiple of 64:
  row phases are taken from -s8)  s9 pair rows the wave flushes (first in bits 4:0, number in bits 11:8) | LDS base of the scan group's accumulators (a multiple of 4096)
  s[10:11] second shifts of a segment pair / scratch  s[14:15] the task (input)
  segment sets A-D: s[16:31] s[32:47] s[48:63] s[64:79]   fetch records A-D (two pair offsets, commit word, turn header):
  s[80:83] s[84:87] s[88:91] s[92:95]   s[96:99] the image's descriptor with the base moved to the pair being loaded
  (s[100:101], s[12:13]: free since format v4)
  v[1:18] / v[19:36] column counters of word 0 / 1: planes p0..p12, pending carries c0..c4
  v37 LDS address of the turn's row-count slots  v38 odd row's count / fourth aligned window  v39 row count
  v40 / v41 odd row's words  v42 a carry  v43 free  v[44:59] four landing sets of four entries  v[60:123] ring
  v124 zero (ring register 64: white runs)  v[125:127] aligned windows, and the carries of the carry-save step between
  words (v125 = ring register 65: dummy commits)
Invariants of the wave program (checked by lint_scalar_loads() below for every variant this file can emit; the memory fault
of round 4 -- profiles/r04_lanes_ablation.md -- was a violation of the second one):
  * scalar loads return OUT OF ORDER, so the only usable wait is lgkmcnt(0) = everything outstanding (the row-count ds_add of
    an odd row counts too).  Between an s_load and the next lgkmcnt(0) its destination SGPRs are IN FLIGHT: nothing may read
    them, write them, or load into them again;
  * which sets are in flight where:  L<cls>_loop (top of a turn): A, B (rows 0, 1 of the turn: requested by row 14 of the
    previous turn, or by the prologue); C, D hold the consumed records of rows 14, 15 -- s[48:57] are free there, which the
    L2 prefetch (a buffer descriptor in s[48:51]) and the flush (s[48:57]) use.  Every even row r starts with lgkmcnt(0) --
    nothing in flight -- and then requests rows r + 2, r + 3 into sets (r + 2) % 4, (r + 3) % 4, which rows r, r + 1 do not
    touch.  The flush ends on its own lgkmcnt(0).  After the last turn A, B are still in flight (two records past the
    stream's end: the program buffer has slack for them): the epilogue waits before it loads the dump pointer into s[16:17];
  * vector loads (the ring's fetches, the prefetch) return IN ORDER, four fetches per row into landing set r % 4, committed
    four rows later behind a COUNTED wait, vmcnt(12); extra loads among the younger ones (the prefetch: two per turn of the
    strip's first scan group) only make that wait stricter.  Row-count atomics are stores: they may retire out of order with
    respect to loads, which can only lengthen a counted wait, never satisfy it early;
  * M0: every word leaves M0 = 0 behind its indexed reads (s_mov m0, 0 in front of the funnel shifts), so every vector-ALU
    instruction outside word() and commit_and_fetch() runs unindexed; vector MEMORY instructions ignore the index mode;
  * row phases (LDS slot, carry-save level, flush) are taken from -s8 modulo 64: the record numbers end on a multiple of 64
    (not of 128); the first record a wave sweeps may have any number that is a multiple of the turn (the records before it --
    virtual rows -- are not in the streams: all registers start at zero, as they would be behind rows that count nothing).
Usage: python tools/gen_slane_asm.py   (writes the .inc; the build only reads it)"""
import os

BLOCK = int(os.environ.get("SLANE_BLOCK", "64"))  # rows between two meetings of the workgroup (16, 32 or 64): its LDS row-count buffers hold BLOCK / 2 pair rows
LB = {16: 4, 32: 5, 64: 6}[BLOCK]
TURN = int(os.environ.get("SLANE_TURN", "16"))  # rows per turn of the loop (4, 8 or 16): the loop's own scalar work is paid once per turn
# timing probes only -- their RESULTS ARE WRONG, only the time means something (profiles/r05_lanes_ablation.md): norec, hotrec,
# nofetch, hotfetch, hotsgpr, nowait, nocommit, noatomic, nobarrier, nom0, noq, constshift, lessE, prio
ABLATE = os.environ.get("SLANE_ABLATE", "").split(",")
PREFETCH = int(os.environ.get("SLANE_PREFETCH", "1"))  # turns ahead the record streams are pulled into L2 by a vector load (0 = off)
LOADBITS = os.environ.get("SLANE_LOADBITS", "")  # cache-policy bits of the source loads (measured: profiles/r05_lanes_ablation.md)
if LOADBITS:
    LOADBITS = " " + LOADBITS
PFREG = 43                     # its landing register (never read)
RING = 60
T0 = 44
NPL = 13                       # counter planes per word: column counts up to 8191
XR = (125, 126, 127, 38)       # aligned windows of a group of four segments (live inside word() only)
DODD = (40, 41)
CNT, CNT2 = 39, 38
LADDR = 37
CARRY = ((125, 126), (127, 42))  # per word: tA, tB -- live inside carry_save() and flush() only, where no aligned window is
P = (1, 19)                    # planes p0..p12 of word k
ST = (14, 32)                  # pending carries c0..c4 of word k
NST = 5
NDUMP = 2 * NPL                # registers dumped: the planes of either word (the parked carries are spent: the number of
                               # rows is a multiple of 64)
SEG = (16, 32, 48, 64)         # segment sets A..D
FOFF = (80, 84, 88, 92)        # fetch records of sets A..D: two pair offsets, the commit word, the turn header
NLOAD = 4                      # loads per row
RSRC = 96                      # s[96:99]: the image's descriptor, its base moved to the pair being loaded (num_records = the pair)
AHEAD = 4


def words2(out, sset, S, d0, d1, E, t):
    """Both words of a row (pk dwords in s[sset + k * S ...], S slots laid out, E of them executed), group by group.  Format v4:
    slot 0 -- the word's first segment -- is read STRAIGHT into the word (the generator names the register pair and the shift
    that put its bits at the top; what lies below is pushed out by the funnel shifts that follow), slots 1 .. 4 are the first
    group of windows, slots 5 .. 7 the second.  The indexed reads of word 0 go to the four window registers, those of word 1 to
    the row's four LANDING registers -- they were committed a moment ago and this row's loads are issued only after the words
    (fetch()) --, so ONE `s_mov m0, 0` per group serves the funnel shifts of both words and the two words' chains are
    independent instruction streams.  Scalar work per slot: one s_lshr writes M0 (ring index + mode bits); the funnel shift
    of slot j rides in bits 25:21 of slot j - 1, and one s_lshr_b64 of the SGPR pair (2 p, 2 p + 1) serves slots 2 p + 1 and
    2 p + 2 (low dword's bits 4:0, high dword's; the junk above bit 4 is ignored by v_alignbit): word 0's land in s[10:11],
    word 1's in vcc."""
    E = E or S
    pk = lambda k, j: "s%d" % (sset + k * S + j)
    X = (XR, (t, t + 1, t + 2, t + 3))
    D = (d0, d1)
    Q = ("s10", "s11"), ("vcc_lo", "vcc_hi")
    NOM0 = "nom0" in ABLATE  # timing probe: what the M0 writes in front of the indexed reads cost (every window reads ring 0 / 1)
    groups = [list(range(0, min(E, 5)))] + ([list(range(5, E))] if E > 5 else [])
    for g, js in enumerate(groups):
        if NOM0:
            out.append("s_mov_b32 m0, 0x3000")
        for k in range(2):
            for j in js:
                if not NOM0:
                    out.append("s_lshr_b32 m0, %s, 5" % pk(k, j))        # ring index + SRC0_REL | SRC1_REL
                shift = "5" if "constshift" in ABLATE else pk(k, j)  # (timing probe: what the SGPR operand of a vector instruction costs)
                if j == 0:   # (index - 1, index) -> the word
                    out.append("v_alignbit_b32 v%d, v%d, v%d, %s" % (D[k], RING, RING - 1, shift))
                else:
                    out.append("v_alignbit_b32 v%d, v%d, v%d, %s" % (X[k][(j - 1) & 3], RING + 1, RING, shift))
        funnels = [j for j in js if j > 0]
        if not funnels:
            continue  # (a word of one segment: M0 is reset by the next indexed write, or by the row's last word below)
        out.append("s_mov_b32 m0, 0")
        for j in funnels:
            for k in range(2):
                if j % 2 == 1 and "noq" not in ABLATE:  # the shifts of slots j, j + 1 sit in slots j - 1, j: an even-aligned SGPR pair
                    out.append("s_lshr_b64 %s, s[%d:%d], 21" % ("s[10:11]" if k == 0 else "vcc", sset + k * S + j - 1, sset + k * S + j))
            for k in range(2):
                out.append("v_alignbit_b32 v%d, v%d, v%d, %s" % (D[k], X[k][(j - 1) & 3], D[k], "7" if "constshift" in ABLATE else Q[k][(j - 1) & 1]))
    if E == 1:
        out.append("s_mov_b32 m0, 0")  # every vector instruction outside the words runs unindexed


def commit(out, x, tset):
    """x = record set of this row; loads return in order: at most (AHEAD - 1) * 4 younger loads may still fly when this
    row's landing set has arrived (row-count atomics in flight only make the wait longer).  The four loads of a row are two
    PAIRS -- two adjacent word columns of one source row in an aligned pair of landing registers -- and a pair is committed
    by ONE v_mov_b64 with DST_REL to an even ring register (format v3: 2 + 2 instructions per row instead of 4 + 4)."""
    t = T0 + 4 * tset
    if "nowait" not in ABLATE:  # (timing probe: what waiting for the landing set costs)
        out.append("s_waitcnt vmcnt(%d)" % ((AHEAD - 1) * NLOAD))
    c0 = FOFF[x] + 2
    for f, ins in enumerate(("s_and_b32 m0, s%d, 0xffff" % c0, "s_lshr_b32 m0, s%d, 16" % c0)):
        if "nocommit" in ABLATE:
            break
        out.append(ins)                                          # even ring register | DST_REL
        out.append("v_mov_b64 v[%d:%d], v[%d:%d]" % (RING, RING + 1, t + 2 * f, t + 2 * f + 1))
    # (M0 is left as it is: the next vector-ALU instruction is a word's first indexed v_alignbit, right behind its own M0 write)


def fetch(out, x, tset):
    """this row's four loads into its landing set, issued BEHIND the row's words (which borrow the landing registers as
    window registers of word 1): they have until the commit four rows on.  Per pair: the descriptor copy's base moves to the
    pair's first entry (operands are read when an instruction is issued: the copy may be rewritten right behind a load), then
    two loads with constant offsets."""
    t = T0 + 4 * tset
    for f in range(2):
        if "nofetch" in ABLATE:
            break
        if "hotsgpr" in ABLATE:  # timing probe (format v3's loads): the offset comes out of an SGPR (s7 = 0x20000), one hot line
            out.append("buffer_load_dword v%d, %%[lane4], s[4:7], s7 offen" % (t + 2 * f))
            out.append("buffer_load_dword v%d, %%[lane4], s[4:7], s7 offen" % (t + 2 * f + 1))
            continue
        if "hotfetch" not in ABLATE:  # (timing probe: every source load reads entries 0 / 1, always in the vector cache)
            out.append("s_add_u32 s%d, s4, s%d" % (RSRC, FOFF[x] + f))
        out.append("buffer_load_dword v%d, %%[lane4], s[%d:%d], 0 offen%s" % (t + 2 * f, RSRC, RSRC + 3, LOADBITS))
        out.append("buffer_load_dword v%d, %%[lane4], s[%d:%d], 0 offen offset:256%s" % (t + 2 * f + 1, RSRC, RSRC + 3, LOADBITS))


def row_count(out, d0, d1, odd, second):
    """Row counts leave the wave as PAIRS: even row in the low half of a dword, odd row in the high half (a count is at
    most 64 per wave and 2480 per row, so neither half can overflow).  The four waves of a scan group in the workgroup --
    four adjacent strips -- add their pairs in LDS (ds_add_u32); every BLOCK rows the workgroup meets and each wave sends
    two of the eight accumulated pair rows on with ONE global atomic per pair row (flush()).  With one global atomic
    per wave and row the kernel issued 1.8e9 L2 atomic requests per launch: 25 ms."""
    if not odd:
        out.append("v_bcnt_u32_b32 v%d, v%d, 0" % (CNT, d0))
        out.append("v_bcnt_u32_b32 v%d, v%d, v%d" % (CNT, d1, CNT))
        return
    out.append("v_bcnt_u32_b32 v%d, v%d, 0" % (CNT2, d0))
    out.append("v_bcnt_u32_b32 v%d, v%d, v%d" % (CNT2, d1, CNT2))
    out.append("v_lshl_or_b32 v%d, v%d, 16, v%d" % (CNT, CNT2, CNT))
    if "noatomic" not in ABLATE:
        out.append("ds_add_u32 v%d, v%d offset:%d" % (LADDR, CNT, 256 * second))  # second: the pair's place in the turn


def flush(out, L):
    """after the last rows of a block of BLOCK rows: the workgroup meets, then the wave reads the pair rows of its scan
    group's LDS buffer that the task assigns to it (s9: first pair row in bits 4:0, their number in bits
    11:8: a workgroup is 4 strips x 4 scan groups, 8 x 2 or 16 x 1, and the BLOCK / 2 pair rows of a scan group's buffer
    are dealt to its strips), clears them and adds them to the candidate's row counts in memory, one pair row per turn of a
    short loop.  Segment set C is dead here: it takes the row-count descriptor, the record count and the pitch (s[48:53])
    and the scratch (s54-s57)."""
    ph = BLOCK - TURN  # first row (mod BLOCK) of the turn that ends a block
    out += ["s_sub_u32 s10, 0, s8", "s_and_b32 s10, s10, %d" % ph, "s_cmp_lg_u32 s10, %d" % ph, "s_cbranch_scc1 %s_nofl" % L]
    if "noatomic" not in ABLATE:
        out += ["s_load_dwordx4 s[48:51], %[desc], 16", "s_load_dwordx2 s[52:53], %[desc], 48",
                "s_waitcnt lgkmcnt(0)"] + ([] if "nobarrier" in ABLATE else ["s_barrier"]) + [
                "s_bfe_u32 s57, s9, 0x40008",                              # pair rows this wave sends on (bits 11:8 = SL_WAVE_COUNT_SHIFT / _BITS, slane.hpp)
                "s_cmp_eq_u32 s57, 0", "s_cbranch_scc1 %s_nofl" % L,
                "s_sub_u32 s54, s52, s8",                                  # row index of the turn (12 mod 16)
                # the block's buffer: that bit of -s8, as the adds computed it (the number of records is a multiple of 64, not
                # of 128: -s8 and the row index agree modulo 64 only), * BLOCK / 2 * 256 bytes
                "s_sub_u32 s55, 0, s8", "s_and_b32 s55, s55, %d" % BLOCK, "s_lshl_b32 s55, s55, 7",
                "s_and_b32 s56, s9, 0xfffff000", "s_add_u32 s55, s55, s56",  # + this scan group's accumulators
                "s_and_b32 s56, s9, 31", "s_lshl_b32 s56, s56, 8", "s_add_u32 s55, s55, s56",  # + first pair row * 256 bytes
                "v_add_u32 v%d, s55, %%[lane4]" % CNT2,
                "v_mov_b32 v%d, 0" % CARRY[1][0],
                "s_lshr_b32 s54, s54, %d" % LB, "s_lshl_b32 s54, s54, %d" % (LB - 1),  # first pair row of the block
                "s_and_b32 s56, s9, 31", "s_add_u32 s54, s54, s56",        # + the wave's first pair row
                "s_mul_i32 s54, s54, s53",                                 # * pitch
                "%s_fl:" % L,
                "ds_read_b32 v%d, v%d" % (CARRY[0][0], CNT2),
                "ds_write_b32 v%d, v%d" % (CNT2, CARRY[1][0]),
                "s_waitcnt lgkmcnt(0)",
                "buffer_atomic_add v%d, %%[lane4], s[48:51], s54 offen" % CARRY[0][0],
                "v_add_u32 v%d, 256, v%d" % (CNT2, CNT2),
                "s_add_u32 s54, s54, s53",
                "s_sub_u32 s57, s57, 1", "s_cmp_lg_u32 s57, 0", "s_cbranch_scc1 %s_fl" % L]
    out.append("%s_nofl:" % L)


def maj(out, d, a, b, c):
    out.append("v_bitop3_b32 v%d, v%d, v%d, v%d bitop3:0xe8" % (d, a, b, c))


def xor3(out, d, a, b, c):
    out.append("v_bitop3_b32 v%d, v%d, v%d, v%d bitop3:0x96" % (d, a, b, c))


def carry_save(out, L, second, grp=0):
    """odd row of a pair: (even row's word in c0, this row's word in DODD) -> the carry-save tree.  First pair of the turn:
    bit 1 of the row index is clear, the carry is parked in c1; second pair: c1 is consumed and bits 2.. of the row index
    (= -rows left, modulo 64) say how far the carry travels."""
    # bits 2.. of the row index say how far the carry travels: the bits below log2(TURN) are known here (grp = this group of
    # four rows within the turn), the others are read from -s8 (the turn's first row, modulo 64).  A carry whose parking
    # level is known when the program is generated is written straight into its parking register (no v_mov).
    known = {4: 0, 8: 1, 16: 2}[TURN]

    def parks_at(lv):  # the carry INTO level lv is parked there (known now)
        return lv == 1 and not second or (second and lv >= 2 and lv - 2 < known and not (grp >> (lv - 2)) & 1)

    for k in range(2):
        maj(out, ST[k] + 1 if parks_at(1) else CARRY[k][0], P[k], ST[k], DODD[k])
        xor3(out, P[k], P[k], ST[k], DODD[k])
    if not second:
        return
    for k in range(2):
        maj(out, ST[k] + 2 if parks_at(2) else CARRY[k][1], P[k] + 1, ST[k] + 1, CARRY[k][0])
        xor3(out, P[k] + 1, P[k] + 1, ST[k] + 1, CARRY[k][0])
    cur = 1
    sub_done = False
    for lv in range(2, NST):
        if lv - 2 < known:
            if not (grp >> (lv - 2)) & 1:  # the carry was parked at this level by the maj above: done
                return
            for k in range(2):
                maj(out, ST[k] + lv + 1 if parks_at(lv + 1) else CARRY[k][cur ^ 1], P[k] + lv, ST[k] + lv, CARRY[k][cur])
                xor3(out, P[k] + lv, P[k] + lv, ST[k] + lv, CARRY[k][cur])
            cur ^= 1
            continue
        if not sub_done:
            out.append("s_sub_u32 s11, 0, s8")
            sub_done = True
        out.append("s_bitcmp1_b32 s11, %d" % lv)
        out.append("s_cbranch_scc1 %s_add%d" % (L, lv))
        for k in range(2):
            out.append("v_mov_b32 v%d, v%d" % (ST[k] + lv, CARRY[k][cur]))
        out.append("s_branch %s_done" % L)
        out.append("%s_add%d:" % (L, lv))
        for k in range(2):
            maj(out, CARRY[k][cur ^ 1], P[k] + lv, ST[k] + lv, CARRY[k][cur])
            xor3(out, P[k] + lv, P[k] + lv, ST[k] + lv, CARRY[k][cur])
        cur ^= 1
    for k in range(2):  # the carry out of the last level ripples into the remaining planes (once per 2^NST rows)
        c = cur
        for lv in range(NST, NPL):
            out.append("v_and_b32 v%d, v%d, v%d" % (CARRY[k][c ^ 1], P[k] + lv, CARRY[k][c]))
            out.append("v_xor_b32 v%d, v%d, v%d" % (P[k] + lv, P[k] + lv, CARRY[k][c]))
            c ^= 1
    out.append("%s_done:" % L)


def rec_loads(out, x, S, row, force=False):
    """request the records of row `row` of the turn (relative to the stream pointers) into set x"""
    if "norec" in ABLATE and not force:
        return
    nd = 2 * S  # segment dwords per row
    out.append("s_load_dwordx%d s[%d:%d], s[0:1], %d" % (nd, SEG[x], SEG[x] + nd - 1, row * nd * 4))
    out.append("s_load_dwordx4 s[%d:%d], s[2:3], %d" % (FOFF[x], FOFF[x] + 3, row * 16))


def body(o, S, L):
    """the row loop of the strips laid out with S slots per word.  A turn = TURN rows; the turn's header (the fetch record of its
    first row, second dword of the commit pair: set A) says how many slots per word its busiest word needs, and the turn
    runs in the loop body that executes exactly that many: S bodies, one dispatch per turn (format v3; the strip's own
    maximum -- round 4 -- executed 0.5 slots per word more)."""
    rec_loads(o, 0, S, 0, True)
    rec_loads(o, 1, S, 1, True)
    if "norec" in ABLATE:
        rec_loads(o, 2, S, 2, True)
        rec_loads(o, 3, S, 3, True)
        o.append("s_waitcnt lgkmcnt(0)")  # (the prefetch below borrows set C's registers)
    o.append("L%s_loop:" % L)
    if PREFETCH:
        # The records of the turn PREFETCH turns ahead, pulled into L2 by two vector loads (a lane per 128-byte line) of the
        # wave of the strip's FIRST scan group (lds_base = 0): a scalar load that misses the scalar cache then finds its
        # line in L2 instead of waiting for HBM.  MUBUF loads through a descriptor made on the spot in s[48:51] (segment set
        # C is dead at the top of a turn: its last records were consumed by rows 14 / 15, the next are requested at row 0),
        # whose num_records is exactly the range wanted: lanes beyond it are dropped by the range check, so no EXEC games
        # and nothing can be read that was not meant to be.  Loads return in order, so the counted vmcnt waits of
        # commit_and_fetch stay correct (two more loads among the younger ones only make them stricter for a few rows);
        # v43 is never read.
        seg_b, fet_b = TURN * 2 * S * 4, TURN * 16
        assert PREFETCH * seg_b < 4096, "the prefetch distance must fit the load's 12-bit offset"
        o += ["s_and_b32 s10, s9, 0xfffff000", "s_cmp_lg_u32 s10, 0", "s_cbranch_scc1 L%s_nopf" % L,
              "s_mov_b32 s48, s0", "s_and_b32 s49, s1, 0xffff", "s_mov_b32 s50, %d" % ((PREFETCH + 1) * seg_b + 64),
              "s_mov_b32 s51, 0x00020000",
              "v_lshlrev_b32 v42, 5, %[lane4]",                                          # lane * 128
              "s_nop 0",
              "buffer_load_dword v%d, v42, s[48:51], 0 offen offset:%d" % (PFREG, PREFETCH * seg_b),
              "s_mov_b32 s48, s2", "s_and_b32 s49, s3, 0xffff", "s_mov_b32 s50, %d" % ((PREFETCH + 1) * fet_b + 64),
              "s_nop 0",
              "buffer_load_dword v%d, v42, s[48:51], 0 offen offset:%d" % (PFREG, PREFETCH * fet_b),
              "L%s_nopf:" % L]
    # LDS address of this turn's pair slots: buffer = bit log2(BLOCK) of the row index, slot = the bits below it down to 1
    o += ["s_sub_u32 s10, 0, s8", "s_bfe_u32 s10, s10, 0x%x" % ((LB << 16) | 1), "s_lshl_b32 s10, s10, 8", "s_and_b32 s11, s9, 0xfffff000",
          "s_add_u32 s10, s10, s11", "v_add_u32 v%d, s10, %%[lane4]" % LADDR]
    # rows 0, 1 of the turn (sets A, B; A carries the turn's header) are here; request rows 2, 3; dispatch on the header
    o.append("s_waitcnt lgkmcnt(0)")
    rec_loads(o, 2, S, 2)
    rec_loads(o, 3, S, 3)
    hdr = FOFF[0] + 3
    for E in range(1, S):
        o += ["s_cmp_eq_u32 s%d, %d" % (hdr, E), "s_cbranch_scc1 L%s_e%d" % (L, E)]
    for E in range(S, 0, -1):
        Ex = max(1, E - 1) if "lessE" in ABLATE else E  # timing probe: what one executed slot per word costs
        o.append("L%s_e%d:" % (L, E))
        for r in range(TURN):
            if r % 2 == 0 and r > 0:  # a batch of two rows: everything requested two rows ago is here; request the next two rows
                o.append("s_waitcnt lgkmcnt(0)")
                rec_loads(o, (r + 2) % 4, S, r + 2)
                rec_loads(o, (r + 3) % 4, S, r + 3)
            odd = r & 1
            commit(o, r % 4, r % 4)
            d = (ST[0], ST[1]) if not odd else DODD
            words2(o, SEG[r % 4], S, d[0], d[1], Ex, T0 + 4 * (r % 4))
            fetch(o, r % 4, r % 4)
            row_count(o, d[0], d[1], odd == 1, r // 2)
            if odd:
                carry_save(o, "L%s_e%d_cs%d" % (L, E, r), (r & 2) != 0, r // 4)
        if E > 1:
            o.append("s_branch L%s_tail" % L)
    o.append("L%s_tail:" % L)
    flush(o, "L%s" % L)
    # (hotrec, a timing probe: the stream pointers stand still, every turn re-reads the first turn's records out of the scalar
    # cache -- the same instructions without the records' latency)
    HOT = "hotrec" in ABLATE
    o += ["s_add_u32 s0, s0, %d" % (0 if HOT else TURN * 2 * S * 4), "s_addc_u32 s1, s1, 0",
          "s_add_u32 s2, s2, %d" % (0 if HOT else TURN * 16), "s_addc_u32 s3, s3, 0",
          "s_sub_u32 s8, s8, %d" % TURN, "s_cmp_lg_u32 s8, 0", "s_cbranch_scc1 L%s_loop" % L]


def kernel():
    """ONE statement: descriptor, dispatch on the slot class, the seven row loops, the counter dump.  (Three statements
    behind a C++ branch keep the class live across them, and with nearly every register clobbered that costs a 129th
    VGPR -- a whole wave per SIMD.)"""
    o = []
    U = "%="  # unique label suffix per asm statement
    o += ["s_load_dwordx4 s[0:3], %[desc], 0", "s_load_dwordx4 s[4:7], %[desc], 32", "s_load_dword s8, %[desc], 76",
          "s_load_dword s11, %[desc], 56", "s_load_dword s9, %[desc], 60", "s_load_dword s10, %[desc], 72"]
    for v in range(1, 128):
        o.append("v_mov_b32 v%d, 0" % v)
    o.append("s_waitcnt lgkmcnt(0)")
    o.append("s_or_b32 s9, s9, s10")
    # the descriptor the source loads go through: the image's, its base moved per pair (fetch()), num_records = one pair (a null
    # task's image has none: its loads read zeros without touching memory)
    o += ["s_mov_b32 s%d, s4" % RSRC, "s_mov_b32 s%d, s5" % (RSRC + 1), "s_min_u32 s%d, s6, 512" % (RSRC + 2), "s_mov_b32 s%d, s7" % (RSRC + 3)]
    if "prio" in ABLATE:  # the four waves of a SIMD (the scan groups of a strip: bits 15:14 of the LDS base) at four priorities
        o += ["s_bfe_u32 s10, s9, 0x2000e", "s_cmp_eq_u32 s10, 1", "s_cbranch_scc0 L%s_p1" % U, "s_setprio 1", "L%s_p1:" % U,
              "s_cmp_eq_u32 s10, 2", "s_cbranch_scc0 L%s_p2" % U, "s_setprio 2", "L%s_p2:" % U,
              "s_cmp_eq_u32 s10, 3", "s_cbranch_scc0 L%s_p3" % U, "s_setprio 3", "L%s_p3:" % U]
    o += ["s_set_gpr_idx_on s10, gpr_idx(SRC0)", "s_mov_b32 m0, 0"]  # index mode on for good; M0 = 0: nothing indexed
    # slot classes (slane.hpp) 0 .. 6 -> slots laid out per word 2, 4, 8, 4, 8, 8, 8; how many are EXECUTED is the turn's business
    for cls, S in ((0, 2), (1, 4), (3, 4)):
        o += ["s_cmp_eq_u32 s11, %d" % cls, "s_cbranch_scc1 L%s_s%d" % (U, S)]
    for S in (8, 4, 2):
        o.append("L%s_s%d:" % (U, S))
        body(o, S, "%s_s%d" % (U, S))
        if S != 2:
            o.append("s_branch L%s_dump" % U)
    # ---- dump the plane registers: [word][p0..p12][lane]
    o.append("L%s_dump:" % U)
    o.append("s_set_gpr_idx_off")
    # the last turn has requested two records past the end into sets A, B: they must have landed before s[16:17] takes the
    # dump's address (scalar loads return out of order: a late record would overwrite it)
    o.append("s_waitcnt lgkmcnt(0)")
    o.append("s_load_dwordx2 s[16:17], %[desc], 64")
    o.append("s_waitcnt vmcnt(0) lgkmcnt(0)")
    for i in range(NDUMP):
        if i and i % 16 == 0:
            o += ["s_add_u32 s16, s16, 4096", "s_addc_u32 s17, s17, 0"]
        o.append("global_store_dword %%[lane4], v%d, s[16:17] offset:%d" % (P[i // NPL] + i % NPL, (i % 16) * 256))
    o.append("s_waitcnt vmcnt(0)")
    return o


def lint_scalar_loads(ins):
    """Scalar loads return OUT OF ORDER and the only wait there is is lgkmcnt(0) -- everything outstanding.  So between an
    s_load and the next s_waitcnt lgkmcnt(0) its destination SGPRs are in flight: no instruction may read them (stale or
    half-written), write them (the late load would overwrite the new value: the round-4 fault -- the epilogue loaded the dump
    pointer into s[16:17] while two records were still travelling into s[16:31]) or be a second load into them.  This pass
    walks the generated program once, in program order, and asserts exactly that; at a label the state is the union over the
    edges that reach it (forward branches recorded when they are seen; a backward branch must not carry more in flight than
    the label was first entered with).  It runs for every variant the generator can emit, probes included."""
    import re
    fly, at_label, pending = set(), {}, {}

    def regs(txt):
        out = set()
        txt = txt.replace("%[desc]", "s[14:15]")
        for a, b in re.findall(r"\bs\[(\d+):(\d+)\]", txt):
            out |= set(range(int(a), int(b) + 1))
        for a in re.findall(r"\bs(\d+)\b", re.sub(r"\bs\[\d+:\d+\]", " ", txt)):
            out.add(int(a))
        return out

    for n, line in enumerate(ins):
        line = line.strip()
        if line.endswith(":"):
            lab = line[:-1]
            fly |= pending.pop(lab, set())
            at_label[lab] = set(fly)
            continue
        op, _, rest = line.partition(" ")
        if op == "s_waitcnt":
            if "lgkmcnt(0)" in rest:
                fly.clear()
            continue
        if op.startswith("s_load_dword"):
            dst, _, src = rest.partition(",")
            d, r = regs(dst), regs(src)
            assert not (d & fly), "instruction %d (%s): a second load into SGPRs that are still in flight: %s" % (n, line, sorted(d & fly))
            assert not (r & fly), "instruction %d (%s): address registers in flight: %s" % (n, line, sorted(r & fly))
            fly |= d
            continue
        used = regs(rest)
        assert not (used & fly), "instruction %d (%s) touches SGPRs with a scalar load in flight: %s" % (n, line, sorted(used & fly))
        if op in ("s_branch",) or op.startswith("s_cbranch"):
            lab = rest.strip()
            if lab in at_label:  # backward
                assert fly <= at_label[lab], "instruction %d (%s): the back edge carries loads in flight the label did not start with: %s" % (
                    n, line, sorted(fly - at_label[lab]))
            else:
                pending[lab] = pending.get(lab, set()) | set(fly)
            if op == "s_branch":
                fly = set()  # unreachable until the next label
    assert not pending, "branches to labels that never came: %s" % sorted(pending)


CLOB = ['"memory"', '"scc"', '"vcc"', '"m0"'] + ['"s%d"' % i for i in list(range(0, 14)) + list(range(16, 102))] + \
       ['"v%d"' % i for i in range(1, 128)]

out = ["// GENERATED by tools/gen_slane_asm.py -- do not edit; see that file for the register map.\n"]
lint_scalar_loads(kernel())
# the lint must see the fault of round 4: the same program without the wait in front of the epilogue's load
_k = kernel()
_i = max(i for i, x in enumerate(_k) if x.startswith("s_load_dwordx2 s[16:17]"))
assert _k[_i - 1] == "s_waitcnt lgkmcnt(0)"
try:
    if "norec" in ABLATE:  # (that probe requests no records in the loop: nothing is in flight at the epilogue)
        raise AssertionError
    lint_scalar_loads(_k[:_i - 1] + _k[_i:])
    raise SystemExit("lint_scalar_loads did not notice a load into s[16:17] under records in flight")
except AssertionError:
    pass
body_txt = "\\n\\t\"\n    \"".join(kernel())
out.append("#define SLANE_ASM \\\n    \"%s\\n\\t\"\n" % body_txt.replace("\n", " \\\n"))
out.append("#define SLANE_ASM_CLOBBERS %s\n" % ", ".join(CLOB))
out.append("#define SLANE_ASM_BLOCK %d  // rows between two meetings of the workgroup\n" % BLOCK)
path = os.environ.get("SLANE_ASM_OUT") or os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                       "omr-img-corrector_amd", "csrc", "slane_asm.inc")
open(path, "w").write("".join(out))
print("wrote", path, len(kernel()), "instructions")
