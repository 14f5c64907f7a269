// dp_task_asm.h -- the row loop of one cooperative DP task (segments, 16-column
// panels, back-pointers) as ONE block of gfx950 assembly.
//
// A task of the segmented cooperative kernel has a SIMD to itself.  A lone wave
// issues at most one instruction every four or five cycles whatever its type,
// pays about twenty for a taken branch, and stalls at every wait for LDS: timed
// with the cells taken out, the 65 other instructions of the first version of
// this loop took 570 cycles a row, more than the 112 vector instructions of the
// row's 16 cells (440).  The compiler's version of dp_panel's row came to about
// 150 instructions besides the cells and 10 taken branches.  Hence:
//   * the distances of rows r and r+1 live in two fixed register sets, and the
//     row body exists twice, once per set: no copies; the two bodies also swap
//     the two registers that hold the left neighbour's values of this row and of
//     the previous one, so what is read ahead lands where the next row wants it;
//   * the loop exists in 24 variants, one per combination of the optional steps
//     (left neighbour; plane injection; hand-over to the right through LDS, as
//     a granule or not at all; plane publication): no step is tested for;
//   * every wait first compares a cached counter and leaves the straight path
//     only to poll;
//   * the values of the next row are read ahead from LDS right after the cells,
//     BEFORE the row's own LDS writes, and the wait for them at the end of the
//     row counts those writes (LDS operations of a wave complete in order, and
//     nothing else is outstanding then): s_waitcnt lgkmcnt(<writes after the
//     reads>) waits for the reads and not for the writes;
//   * addresses advance by additions.
// What it computes, stores and announces, and in which order, is dp_panel's
// COOP + GRAN path, statement for statement (see the comments there); the tests
// run every cooperative case through both.
//
// Register map inside the block (all named in the clobber list):
//   s36 / s[40:55]   distances of the even rows: C(i, first cell), C(i, cells 1..16)
//   s37 / s[56:71]   the same for the odd rows
//   s[72:73], s[94:95], s[98:99] address of the first row of distances, + 4, + 68 bytes    s93 offset of the
//   next row to load    s74 row r    s79 r + 1    s[96:97] exec mask of lane 63
//   s76, s78 the rows up to which the right neighbour's ring / the out ring have room (last seen counter
//   + 7 / + 62 + 15)    s75, s77 last polled counters of the left neighbour and the plane fetch wave
//   s80..s82, s86, s87 scratch (s82: polls of the wait at hand)   s[84:85] = 512   s92 = min(rows, 62)
//   s88..s91 diagnostic build: how often each wait left the straight path
//   v[64:79] the row of costs   v96 v97 diagonal sources   v98 t   v99 back-pointer bits
//   v100 / v101 the left neighbour's value of this row and of the previous one (= "one up, one left" of
//   cell 0; lane 0: the plane's boundary value); the two rows of the loop swap the two registers
//   v102 v103 scratch   v104 / v105 the left neighbour's / the plane fetch wave's counter, read ahead   v[106:107] granule
#pragma once

namespace dp {

#define T16_ONE(MPI, MPO, CN, RJ, RJM1)                                                \
  "v_add_f32 " MPO ", " CN ", " RJ "\n\t"                                              \
  "v_min_f32_dpp v98, " MPI ", " RJ " wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t" \
  "v_cmp_eq_f32 vcc, v98, " RJ "\n\t"                                                  \
  "v_addc_co_u32 v99, vcc, v99, v99, vcc\n\t"                                          \
  "v_min_f32 " RJ ", v98, " RJM1 "\n\t"                                                \
  "v_cmp_eq_f32 vcc, " RJ ", " RJM1 "\n\t"                                             \
  "v_addc_co_u32 v99, vcc, v99, v99, vcc\n\t"

// sixteen cells: R0..R15 their row registers, LEFT the value left of the first
#define T16_CELLS_(LEFT, R0, R1, R2, R3, R4, R5, R6, R7, R8, R9, R10, R11, R12, R13, R14, R15, \
                   C1, C2, C3, C4, C5, C6, C7, C8, C9, C10, C11, C12, C13, C14, C15, C16)  \
  T16_ONE("v96", "v97", C1, R0, LEFT)                                                  \
  T16_ONE("v97", "v96", C2, R1, R0)                                                    \
  T16_ONE("v96", "v97", C3, R2, R1)                                                    \
  T16_ONE("v97", "v96", C4, R3, R2)                                                    \
  T16_ONE("v96", "v97", C5, R4, R3)                                                    \
  T16_ONE("v97", "v96", C6, R5, R4)                                                    \
  T16_ONE("v96", "v97", C7, R6, R5)                                                    \
  T16_ONE("v97", "v96", C8, R7, R6)                                                    \
  T16_ONE("v96", "v97", C9, R8, R7)                                                    \
  T16_ONE("v97", "v96", C10, R9, R8)                                                   \
  T16_ONE("v96", "v97", C11, R10, R9)                                                  \
  T16_ONE("v97", "v96", C12, R11, R10)                                                 \
  T16_ONE("v96", "v97", C13, R12, R11)                                                 \
  T16_ONE("v97", "v96", C14, R13, R12)                                                 \
  T16_ONE("v96", "v97", C15, R14, R13)                                                 \
  T16_ONE("v97", "v96", C16, R15, R14)
#define T16_CELLS(LIN, ...)                                                            \
  T16_CELLS_(LIN, "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", \
             "v78", "v79", __VA_ARGS__)
#define T16_CELLS_HI(...)                                                              \
  T16_CELLS_("v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", \
             "v94", "v95", __VA_ARGS__)

// coop_wait_cached: poll the counter at LDS address PADDR until it reaches NEED
// (kept in KNOWN); eight polls back to back, then the abort flag is consulted
// and the wave sleeps between polls; after %[spin] polls the wait gives up for
// the whole workgroup.  NEED and KNOWN must not be s81, s82.
#define T16_POLL(TAG, PADDR, NEED, KNOWN)                                              \
  "s_mov_b32 s82, 0\n"                                                                 \
  "Lpl_" TAG "_%=:\n\t"                                                                \
  "ds_read_b32 v102, " PADDR "\n\t"                                                     \
  "s_waitcnt lgkmcnt(0)\n\t"                                                           \
  "v_readfirstlane_b32 " KNOWN ", v102\n\t"                                             \
  "s_cmp_ge_i32 " KNOWN ", " NEED "\n\t"                                               \
  "s_cbranch_scc1 Lpd_" TAG "_%=\n\t"                                                  \
  "s_add_i32 s82, s82, 1\n\t"                                                          \
  "s_cmp_lt_i32 s82, 8\n\t"                                                            \
  "s_cbranch_scc1 Lpl_" TAG "_%=\n\t"                                                  \
  "ds_read_b32 v102, %[pabort]\n\t"                                                     \
  "s_waitcnt lgkmcnt(0)\n\t"                                                           \
  "v_readfirstlane_b32 s81, v102\n\t"                                                   \
  "s_cmp_lg_u32 s81, 0\n\t"                                                            \
  "s_cbranch_scc1 Lpd_" TAG "_%=\n\t"                                                  \
  "s_sleep 1\n\t"                                                                      \
  "s_cmp_gt_i32 s82, %[spin]\n\t"                                                      \
  "s_cbranch_scc0 Lpl_" TAG "_%=\n\t"                                                  \
  "v_mov_b32 v102, 1\n\t"                                                               \
  "v_mov_b32 v103, 0\n\t"                                                               \
  "ds_write_b32 %[pabort], v102\n\t"                                                    \
  "global_store_dword v103, v102, %[errp]\n\t"                                           \
  "s_branch Lpl_" TAG "_%=\n"                                                          \
  "Lpd_" TAG "_%=:\n\t"

// lane 0: the plane row at LDS address v103 -> boundary value into BND and the row registers
// (T16_W_INJECT_MORE: the registers of the columns from 8 on)
#define T16_INJECT_READS(BND)                                                          \
  "s_mov_b64 exec, 1\n\t"                                                              \
  "ds_read_b32 " BND ", v103\n\t"                                                      \
  "ds_read2_b32 v[64:65], v103 offset0:1 offset1:2\n\t"                                \
  "ds_read2_b32 v[66:67], v103 offset0:3 offset1:4\n\t"                                \
  "ds_read2_b32 v[68:69], v103 offset0:5 offset1:6\n\t"                                \
  "ds_read2_b32 v[70:71], v103 offset0:7 offset1:8\n\t"                                \
  T16_W_INJECT_MORE                                                                    \
  "s_mov_b64 exec, -1\n\t"

#define T16_CAT_(A, B) A##B
#define T16_SEL(NAME, V) T16_CAT_(NAME, V)

// ---- the optional steps; the digit is the variant's: 0 = the task does not have the step

// top of a row: is the left neighbour's value (read ahead into LIN) there?
#define T16_TOPLEFT_0(X)
#define T16_TOPLEFT_1(X)                                                               \
  "v_cmp_le_i32 vcc, s79, v104\n\t"                                                    \
  "s_cbranch_vccz Lleft_slow_" X "_%=\n"                                                \
  "Lleft_done_" X "_%=:\n\t"
// ... and the row of the previous block's last layer (read ahead into lane 0)?
#define T16_TOPINJ_0(X)
#define T16_TOPINJ_1(X)                                                                \
  "v_cmp_le_i32 vcc, s79, v105\n\t"                                                    \
  "s_cbranch_vccz Linj_slow_" X "_%=\n"                                                 \
  "Linj_done_" X "_%=:\n\t"
// after the cells: read ahead for the next row, counter first, data after it.  The next row's LIN
// is this row's PIN, which the cells are done with; its PIN is this row's LIN.  (After the last row
// the reads fetch a ring slot nobody uses.)
#define T16_AHEADLEFT_0(PIN)
#define T16_AHEADLEFT_1(PIN)                                                           \
  "ds_read_b32 v104, %[pleft]\n\t"                                                      \
  "s_and_b32 s80, s79, 7\n\t"                                                          \
  "s_lshl_b32 s80, s80, 8\n\t"                                                         \
  "v_add_u32 v102, s80, %[rin]\n\t"                                                     \
  "ds_read_b32 " PIN ", v102\n\t"
#define T16_AHEADINJ_0(LIN)
#define T16_AHEADINJ_1(LIN)                                                            \
  "ds_read_b32 v105, %[pplane]\n\t"                                                     \
  "s_and_b32 s80, s79, 15\n\t"                                                         \
  "s_mul_i32 s80, s80, %[prow]\n\t"                                                    \
  "s_add_i32 s80, s80, %[pring]\n\t"                                                   \
  "v_mov_b32 v103, s80\n\t"                                                             \
  T16_INJECT_READS(LIN)
// the last column to the right neighbour: 1 = LDS ring slot r % 8 (free once the neighbour has
// finished row r - 8), 2 = to another workgroup, one write-through store of {value, tag}
#define T16_RIGHT_0(X)
#define T16_RIGHT_1(X)                                                                 \
  "s_cmp_ge_i32 s76, s74\n\t"                 /* s76: the neighbour's last seen row + 7 */ \
  "s_cbranch_scc0 Lring_slow_" X "_%=\n"                                                \
  "Lring_go_" X "_%=:\n\t"                                                             \
  "s_and_b32 s81, s74, 7\n\t"                                                          \
  "s_lshl_b32 s81, s81, 8\n\t"                                                         \
  "v_add_u32 v102, s81, %[rout]\n\t"                                                    \
  "ds_write_b32 v102, " T16_W_LASTCOL "\n\t"
#define T16_RIGHT_2(X)                                                                 \
  "s_add_i32 s80, s79, %[gbase]\n\t"                                                   \
  "v_mov_b32 v106, " T16_W_LASTCOL "\n\t"                                                             \
  "v_mov_b32 v107, s80\n\t"                                                             \
  "global_store_dwordx2 %[xo], v[106:107], off sc1\n\t"                                  \
  "v_lshl_add_u64 %[xo], %[xo], 0, s[84:85]\n\t"
// LDS writes of a row after its read-ahead, without / with a published plane row
#define T16_NWRITES_0 "1"
#define T16_NWRITES_1 "2"
#define T16_NWRITES_2 "1"
// (with a published row: T16_W_NWRITESPUB_<R>, four or eight 16-byte writes more)
// end of a row: announce it (lane 0 stores r + 1 behind the row's LDS writes), wait for what was
// read ahead, go to the next row
#define T16_ROWEND(NX, NW, LIM, DONE)                                                  \
  "s_mov_b64 exec, 1\n\t"                                                              \
  "ds_write_b32 %[pmy], v103\n\t"                                                       \
  "s_mov_b64 exec, -1\n\t"                                                             \
  T16_ROWNEXT(NX, NW, LIM, DONE)
#define T16_ROWNEXT(NX, NW, LIM, DONE)                                                 \
  "s_waitcnt lgkmcnt(" NW ")\n\t"                                                      \
  "s_mov_b32 s74, s79\n\t"                                                             \
  "s_cmp_lt_i32 s74, " LIM "\n\t"                                                      \
  "s_cbranch_scc1 Lrow_" NX "_%=\n\t"                                                  \
  "s_branch " DONE "_%=\n"
// lane 63's row = row r - 62 of the next block's plane: into out-ring slot r % 16
// (MINRMS_COOP_ORING), free once the publish wave has stored row r - 62 - 16
#define T16_TAIL_0(X, NX, R, LIM, DONE)                                                \
  T16_ROWEND(NX, T16_SEL(T16_NWRITES_, R), LIM, DONE)
#define T16_TAIL_1(X, NX, R, LIM, DONE)                                                \
  "s_cmp_ge_i32 s78, s74\n\t"                 /* s78: the publish wave's last seen row + 62 + 15 */ \
  "s_cbranch_scc0 Lpub_slow_" X "_%=\n"                                                 \
  "Lpub_go_" X "_%=:\n\t"                                                              \
  "s_and_b32 s81, s74, 15\n\t"                                                         \
  "s_lshl_b32 s81, s81, " T16_W_SLOTSHIFT "\n\t"                                       \
  "v_add_u32 v102, s81, %[oring]\n\t"                                                  \
  "s_mov_b64 exec, s[96:97]\n\t"              /* lane 63, which also announces the row */ \
  "ds_write_b128 v102, v[64:67]\n\t"                                                   \
  "ds_write_b128 v102, v[68:71] offset:16\n\t"                                         \
  T16_W_PUBWRITES_MORE                                                                 \
  "ds_write_b32 %[pmy], v103\n\t"                                                       \
  "s_mov_b64 exec, -1\n\t"                                                             \
  T16_ROWNEXT(NX, T16_SEL(T16_W_NWRITESPUB_, R), LIM, DONE)

// ---- what depends on the panel width; these are the definitions for 16 columns (those for 32
// follow the first task function below)

// distances and cells of a row.  16 columns: the row's distances are in one register set (CUR0,
// C1..C16) and the next row's are loaded into the other (NXT0, NXTV) while the cells run; s[72:73]
// points at the next row to load.
#define T16_MIDDLE_16(AHEAD, LIN, PIN, CUR0, NXT0, NXTV, ...)                          \
  "v_add_f32 v96, " CUR0 ", " PIN "\n\t"      /* three instructions before the cells read it through DPP */ \
  "s_load_dword " NXT0 ", s[72:73], s93\n\t"                                           \
  "s_load_dwordx16 " NXTV ", s[94:95], s93\n\t"                                        \
  "s_add_i32 s93, s93, %[cstr]\n\t"           /* after the last row: the row behind the table (padding), not used */ \
  T16_CELLS(LIN, __VA_ARGS__)                                                          \
  "global_store_dword %[bp], v99, off\n\t"                                             \
  "s_waitcnt lgkmcnt(0)\n\t"                   /* the next row's distances are there; nothing is outstanding */ \
  AHEAD
// 8 columns: as for 16, with half the distances; the eight codes of a row are half a back-pointer
// word (bp_addr points at the half)
#define T16_MIDDLE_8(AHEAD, LIN, PIN, CUR0, NXT0, NXTV, C1, C2, C3, C4, C5, C6, C7, C8, ...) \
  "v_add_f32 v96, " CUR0 ", " PIN "\n\t"                                               \
  "s_load_dword " NXT0 ", s[72:73], s93\n\t"                                           \
  "s_load_dwordx8 " NXTV ", s[94:95], s93\n\t"                                         \
  "s_add_i32 s93, s93, %[cstr]\n\t"                                                    \
  T16_ONE("v96", "v97", C1, "v64", LIN)                                                \
  T16_ONE("v97", "v96", C2, "v65", "v64")                                              \
  T16_ONE("v96", "v97", C3, "v66", "v65")                                              \
  T16_ONE("v97", "v96", C4, "v67", "v66")                                              \
  T16_ONE("v96", "v97", C5, "v68", "v67")                                              \
  T16_ONE("v97", "v96", C6, "v69", "v68")                                              \
  T16_ONE("v96", "v97", C7, "v70", "v69")                                              \
  T16_ONE("v97", "v96", C8, "v71", "v70")                                              \
  "global_store_short %[bp], v99, off\n\t"                                             \
  "s_waitcnt lgkmcnt(0)\n\t"                   /* the next row's distances are there; nothing is outstanding */ \
  AHEAD                                        /* (asked for four cells earlier, the wait for the distances would come too early: slower) */
// 32 columns: s36 / s[40:55] hold the row's first distance and those of cells 1..16 at the top of
// every row, s[56:71] get those of cells 17..32 while the first sixteen cells run, and the next
// row's first half is loaded while the second sixteen run; s93 is the offset of the current row.
#define T16_MIDDLE_32(AHEAD, LIN, PIN, CUR0, NXT0, NXTV, ...)                          \
  "v_add_f32 v96, s36, " PIN "\n\t"                                                    \
  "s_load_dwordx16 s[56:71], s[98:99], s93\n\t"                                        \
  "s_add_i32 s93, s93, %[cstr]\n\t"           /* after the last row: the row behind the table (padding), not used */ \
  "s_nop 0\n\t"                                                                        \
  T16_CELLS(LIN, "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55") \
  "global_store_dword %[bp], v99, off\n\t"                                             \
  "s_waitcnt lgkmcnt(0)\n\t"                   /* the second half of this row's distances */ \
  "s_load_dword s36, s[72:73], s93\n\t"                                                \
  "s_load_dwordx16 s[40:55], s[94:95], s93\n\t"                                        \
  T16_CELLS_HI("s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71") \
  "global_store_dword %[bp], v99, off offset:256\n\t"                                  \
  "s_waitcnt lgkmcnt(0)\n\t"                   /* the next row's first half; nothing is outstanding */ \
  AHEAD

#define T16_W_MIDDLE(...) T16_MIDDLE_16(__VA_ARGS__)
#define T16_W_SETA T16_SETA_16
#define T16_W_SETB T16_SETB_16
#define T16_W_FIRST_LOAD "s_load_dwordx16 s[40:55], s[72:73], 0x4\n\t"
#define T16_W_FIRST_ADVANCE "s_mov_b32 s93, %[cstr]\n\t"   /* 16 columns: row 1 is the next to load */
#define T16_W_INIT_MORE
#define T16_W_PICKS_MORE
#define T16_W_LASTCOL "v79"
#define T16_INJECT_8TO15                                                               \
  "ds_read2_b32 v[72:73], v103 offset0:9 offset1:10\n\t"                               \
  "ds_read2_b32 v[74:75], v103 offset0:11 offset1:12\n\t"                              \
  "ds_read2_b32 v[76:77], v103 offset0:13 offset1:14\n\t"                              \
  "ds_read2_b32 v[78:79], v103 offset0:15 offset1:16\n\t"
#define T16_PUBWRITES_8TO15                                                            \
  "ds_write_b128 v102, v[72:75] offset:32\n\t"                                         \
  "ds_write_b128 v102, v[76:79] offset:48\n\t"
#define T16_W_INJECT_MORE T16_INJECT_8TO15
#define T16_W_PUBWRITES_MORE T16_PUBWRITES_8TO15
#define T16_W_SLOTSHIFT "6"
#define T16_W_NWRITESPUB_0 "5"
#define T16_W_NWRITESPUB_1 "6"
#define T16_W_NWRITESPUB_2 "5"

// One row.  L, I, R: the variant (left neighbour, injection, right hand-over 0/1/2); T: 1 = the row
// publishes (rows from 62 on of a task that publishes).  X: the row's label, NX: the label of the row
// that follows while the row number is below LIM; then to DONE.  CUR0, C1..C16: this row's distances; NXT0, NXTV: where the
// next row's go.  LIN: the register that holds the left neighbour's value of this row, PIN: of the
// previous row.
#define T16_ROW(L, I, R, T, X, NX, LIM, DONE, LIN, PIN, CUR0, NXT0, NXTV, C1, C2, C3, C4, C5, C6, C7, C8, C9, C10, C11, C12, C13, C14, C15, C16) \
  "Lrow_" X "_%=:\n\t"                                                                 \
  "s_add_i32 s79, s74, 1\n\t"                                                          \
  T16_SEL(T16_TOPLEFT_, L)(X)                                                          \
  T16_SEL(T16_TOPINJ_, I)(X)                                                           \
  T16_W_MIDDLE(T16_SEL(T16_AHEADLEFT_, L)(PIN), LIN, PIN, CUR0, NXT0, NXTV, C1, C2, C3, C4, C5, C6, C7, C8, C9, C10, C11, C12, C13, C14, C15, C16) \
  T16_SEL(T16_AHEADINJ_, I)(LIN)                                                       \
  "v_lshl_add_u64 %[bp], %[bp], 0, %[bpstr]\n\t"                                       \
  T16_SEL(T16_RIGHT_, R)(X)                                                            \
  "v_mov_b32 v103, s79\n\t"                                                             \
  T16_SEL(T16_TAIL_, T)(X, NX, R, LIM, DONE)

// The waits that did not find their counter far enough, out of the straight path.
// (Diagnostic build: s88..s91 count how often each is taken.)
#ifdef MINRMS_COOP_STAMPS
#define T16_COUNT(R) "s_add_i32 " R ", " R ", 1\n\t"
#else
#define T16_COUNT(R)
#endif
#define T16_SLOWLEFT_0(X, LIN)
#define T16_SLOWLEFT_1(X, LIN)                                                         \
  "Lleft_slow_" X "_%=:\n\t"                                                           \
  T16_COUNT("s88")                                                                     \
  T16_POLL("left" X, "%[pleft]", "s79", "s75")                                         \
  "s_and_b32 s80, s74, 7\n\t"                                                          \
  "s_lshl_b32 s80, s80, 8\n\t"                                                         \
  "v_add_u32 v102, s80, %[rin]\n\t"                                                     \
  "ds_read_b32 " LIN ", v102\n\t"                                                       \
  "s_waitcnt lgkmcnt(0)\n\t"                                                           \
  "s_branch Lleft_done_" X "_%=\n"
#define T16_SLOWINJ_0(X, PIN)
#define T16_SLOWINJ_1(X, PIN)                                                          \
  "Linj_slow_" X "_%=:\n\t"                                                            \
  T16_COUNT("s89")                                                                     \
  T16_POLL("inj" X, "%[pplane]", "s79", "s77")                                         \
  "s_and_b32 s80, s74, 15\n\t"                                                         \
  "s_mul_i32 s80, s80, %[prow]\n\t"                                                    \
  "s_add_i32 s80, s80, %[pring]\n\t"                                                   \
  "v_mov_b32 v103, s80\n\t"                                                             \
  T16_INJECT_READS(PIN)                                                                \
  "s_waitcnt lgkmcnt(0)\n\t"                                                           \
  "s_branch Linj_done_" X "_%=\n"
#define T16_SLOWRING_0(X)
#define T16_SLOWRING_2(X)
#define T16_SLOWRING_1(X)                                                              \
  "Lring_slow_" X "_%=:\n\t"                                                           \
  T16_COUNT("s90")                                                                     \
  "s_add_i32 s80, s74, -7\n\t"                                                         \
  T16_POLL("ring" X, "%[pright]", "s80", "s76")                                        \
  "s_add_i32 s76, s76, 7\n\t"                                                          \
  "s_branch Lring_go_" X "_%=\n"
#define T16_SLOWPUB_0(X)
#define T16_SLOWPUB_1(X)                                                               \
  "Lpub_slow_" X "_%=:\n\t"                                                            \
  T16_COUNT("s91")                                                                     \
  "s_add_i32 s86, s74, -77\n\t"                                                        \
  T16_POLL("pub" X, "%[ppub]", "s86", "s78")                                           \
  "s_add_i32 s78, s78, 77\n\t"                                                         \
  "v_mov_b32 v103, s79\n\t"                     /* the poll may have used it */         \
  "s_branch Lpub_go_" X "_%=\n"
#define T16_SLOW(L, I, R, P, X, LIN, PIN)                                              \
  T16_SEL(T16_SLOWLEFT_, L)(X, LIN)                                                    \
  T16_SEL(T16_SLOWINJ_, I)(X, PIN)                                                     \
  T16_SEL(T16_SLOWRING_, R)(X)                                                         \
  T16_SEL(T16_SLOWPUB_, P)(X)

struct TaskAsmArgs {
  const float *crow;        // C(first row of the block, first cell of the panel)
  uint32_t cstride_bytes;   // one row of the distance table
  int32_t nrows;
  uint32_t flags;           // 1 left neighbour, 2 plane injection, 4 right neighbour through the LDS ring, 8 ... as granules, 16 publishes plane rows
  uint64_t bp_addr;         // this lane's back-pointer word of row 0
  uint64_t bp_stride_bytes;
  uint64_t xout_addr;       // this lane's granule of row 0 (flags & 8)
  int32_t gbase;
  uint32_t p_my, p_left, p_right, p_plane, p_pub, p_abort;   // LDS byte addresses of the counters
  uint32_t ring_in, ring_out;   // LDS byte addresses: this lane's slot 0 of the left neighbour's ring, of the own ring
  uint32_t pring_at;        // LDS byte address of "one left of cell 0" in row 0 of the plane ring
  uint32_t pring_row_bytes;
  uint32_t oring_at;        // LDS byte address of this task's slot 0 of the out ring
  float base_init;
  int *err;
  int32_t spin_limit;
  int32_t jl;               // column of the answer cell inside the panel, or outside 0..15
};

// a wave-uniform value as the compiler must see it to keep it in scalar registers
__device__ __forceinline__ uint32_t t16_uniform(const uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t t16_uniform64(const uint64_t v)
{
  return ((uint64_t)t16_uniform((uint32_t)(v >> 32)) << 32) | t16_uniform((uint32_t)v);
}

#ifdef MINRMS_COOP_STAMPS
#define T16_DBG_INIT "s_mov_b32 s88, 0\n\ts_mov_b32 s89, 0\n\ts_mov_b32 s90, 0\n\ts_mov_b32 s91, 0\n\t"
#define T16_DBG_OUT                                                                    \
  "v_mov_b32 %[dbg], s88\n\t"                                                          \
  "v_lshl_add_u32 %[dbg], s89, 8, %[dbg]\n\t"                                          \
  "v_lshl_add_u32 %[dbg], s90, 16, %[dbg]\n\t"                                         \
  "v_lshl_add_u32 %[dbg], s91, 24, %[dbg]\n\t"
#define T16_DBG_OPERAND , [dbg] "=&v"(dbg)
#else
#define T16_DBG_INIT
#define T16_DBG_OUT
#define T16_DBG_OPERAND
#endif

#define T16_PICK(K, V) "s_cmp_eq_u32 %[jl], " K "\n\ts_cselect_b64 vcc, -1, 0\n\tv_cndmask_b32 %[cv], %[cv], " V ", vcc\n\t"

#define T16_SETA_16 "s36", "s37", "s[56:71]", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55"
#define T16_SETB_16 "s37", "s36", "s[40:55]", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71"
#define T16_SETA_8 "s36", "s37", "s[56:63]", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", "s54", "s55"
#define T16_SETB_8 "s37", "s36", "s[40:47]", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71"
#define T16_SETA T16_W_SETA
#define T16_SETB T16_W_SETB
#define T16_ROW_(...) T16_ROW(__VA_ARGS__)
// A task that does not publish: one loop over all rows.
#define T16_PHASES_0(L, I, R)                                                          \
  T16_ROW_(L, I, R, 0, "a", "b", "%[nrows]", "Lrows_done", "v100", "v101", T16_SETA)     \
  T16_ROW_(L, I, R, 0, "b", "a", "%[nrows]", "Lrows_done", "v101", "v100", T16_SETB)
#define T16_SLOWS_0(L, I, R)                                                           \
  T16_SLOW(L, I, R, 0, "a", "v100", "v101")                                              \
  T16_SLOW(L, I, R, 0, "b", "v101", "v100")
// A task that publishes: rows 0..61 (lane 63 is not yet in the next block's plane), then the rest.
// Row 62 is an even row: the second loop starts with the "a" registers like the first.
#define T16_PHASES_1(L, I, R)                                                          \
  "s_min_i32 s92, %[nrows], 62\n\t"                                                    \
  T16_ROW_(L, I, R, 0, "a", "b", "s92", "Lrows62", "v100", "v101", T16_SETA)             \
  T16_ROW_(L, I, R, 0, "b", "a", "s92", "Lrows62", "v101", "v100", T16_SETB)             \
  "Lrows62_%=:\n\t"                                                                    \
  "s_cmp_lt_i32 s74, %[nrows]\n\t"                                                     \
  "s_cbranch_scc0 Lrows_done_%=\n\t"                                                   \
  T16_ROW_(L, I, R, 1, "pa", "pb", "%[nrows]", "Lrows_done", "v100", "v101", T16_SETA)   \
  T16_ROW_(L, I, R, 1, "pb", "pa", "%[nrows]", "Lrows_done", "v101", "v100", T16_SETB)
#define T16_SLOWS_1(L, I, R)                                                           \
  T16_SLOW(L, I, R, 0, "a", "v100", "v101")                                              \
  T16_SLOW(L, I, R, 0, "b", "v101", "v100")                                              \
  T16_SLOW(L, I, R, 1, "pa", "v100", "v101")                                             \
  T16_SLOW(L, I, R, 1, "pb", "v101", "v100")

// The whole task of one variant: set-up, the two row bodies, the answer cell, the slow paths.
#define T16_TASK(L, I, R, P)                                                           \
  asm volatile(                                                                        \
      "s_mov_b64 s[72:73], %[crow]\n\t"                                                \
      "s_load_dword s36, s[72:73], 0x0\n\t"                                            \
      T16_W_FIRST_LOAD                                                                 \
      "s_add_u32 s94, s72, 4\n\t"                                                      \
      "s_addc_u32 s95, s73, 0\n\t"                                                     \
      "s_add_u32 s98, s72, 0x44\n\t"                                                   \
      "s_addc_u32 s99, s73, 0\n\t"                                                     \
      T16_W_FIRST_ADVANCE                                                              \
      "s_mov_b32 s96, 0\n\t"                                                           \
      "s_mov_b32 s97, 0x80000000\n\t"                                                  \
      "s_mov_b32 s74, 0\n\t"                                                           \
      "s_mov_b32 s75, 0\n\t"                                                           \
      "s_mov_b32 s76, 7\n\t"                                                           \
      "s_mov_b32 s77, 0\n\t"                                                           \
      "s_mov_b32 s78, 77\n\t"                                                          \
      "s_mov_b64 s[84:85], 0x200\n\t"                                                  \
      T16_DBG_INIT                                                                     \
      "v_mov_b32 v64, %[binit]\n\t"                                                    \
      "v_mov_b32 v65, %[binit]\n\t"                                                    \
      "v_mov_b32 v66, %[binit]\n\t"                                                    \
      "v_mov_b32 v67, %[binit]\n\t"                                                    \
      "v_mov_b32 v68, %[binit]\n\t"                                                    \
      "v_mov_b32 v69, %[binit]\n\t"                                                    \
      "v_mov_b32 v70, %[binit]\n\t"                                                    \
      "v_mov_b32 v71, %[binit]\n\t"                                                    \
      "v_mov_b32 v72, %[binit]\n\t"                                                    \
      "v_mov_b32 v73, %[binit]\n\t"                                                    \
      "v_mov_b32 v74, %[binit]\n\t"                                                    \
      "v_mov_b32 v75, %[binit]\n\t"                                                    \
      "v_mov_b32 v76, %[binit]\n\t"                                                    \
      "v_mov_b32 v77, %[binit]\n\t"                                                    \
      "v_mov_b32 v78, %[binit]\n\t"                                                    \
      "v_mov_b32 v79, %[binit]\n\t"                                                    \
      T16_W_INIT_MORE                                                                  \
      "v_mov_b32 v100, %[binit]\n\t"                                                   \
      "v_mov_b32 v101, %[binit]\n\t"                                                    \
      "v_mov_b32 v104, 0\n\t"                                                           \
      "v_mov_b32 v105, 0\n\t"                                                           \
      "s_waitcnt lgkmcnt(0)\n\t"                 /* the first row's distances */        \
      T16_SEL(T16_PHASES_, P)(L, I, R)                                                 \
      "Lrows_done_%=:\n\t"                                                             \
      "s_waitcnt vmcnt(0) lgkmcnt(0)\n\t"                                              \
      /* cost(n, N1, N2): the last row's value in column jl */                         \
      "v_mov_b32 %[cv], 0\n\t"                                                         \
      T16_PICK("0", "v64") T16_PICK("1", "v65") T16_PICK("2", "v66") T16_PICK("3", "v67") \
      T16_PICK("4", "v68") T16_PICK("5", "v69") T16_PICK("6", "v70") T16_PICK("7", "v71") \
      T16_PICK("8", "v72") T16_PICK("9", "v73") T16_PICK("10", "v74") T16_PICK("11", "v75") \
      T16_PICK("12", "v76") T16_PICK("13", "v77") T16_PICK("14", "v78") T16_PICK("15", "v79") \
      T16_W_PICKS_MORE                                                                 \
      T16_DBG_OUT                                                                      \
      "s_branch Lexit_%=\n"                                                            \
      T16_SEL(T16_SLOWS_, P)(L, I, R)                                                  \
      "Lexit_%=:\n\t"                                                                  \
      : [cv] "=&v"(cv), [bp] "+v"(bp), [xo] "+v"(xo) T16_DBG_OPERAND                    \
      : [crow] "s"(crow), [cstr] "s"(cstr), [nrows] "s"(nrows),                        \
        [bpstr] "s"(bpstr), [gbase] "s"(gbase),                                        \
        [pmy] "v"(a.p_my), [pleft] "v"(a.p_left), [pright] "v"(a.p_right), [pplane] "v"(a.p_plane), \
        [ppub] "v"(a.p_pub), [pabort] "v"(a.p_abort), [rin] "v"(a.ring_in), [rout] "v"(a.ring_out), \
        [pring] "s"(pring), [prow] "s"(prow), [oring] "v"(a.oring_at),                 \
        [binit] "v"(a.base_init), [errp] "s"(errp), [spin] "s"(spin), [jl] "s"(jl)     \
      : "s36", "s37", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51", "s52", "s53", \
        "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", \
        "s70", "s71", "s72", "s73", "s74", "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", \
        "s86", "s87", "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97", "s98", "s99", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", \
        "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", \
        "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", \
        "v106", "v107", "vcc", "scc", "memory")

// flags: 1 left neighbour, 2 injection, 4 right through the ring, 8 right as granules, 16 publication
#define T16_CASE(L, I, R, P) case ((L) | (I) << 1 | ((R) == 1) << 2 | ((R) == 2) << 3 | (P) << 4): T16_TASK(L, I, R, P); break;
#ifdef MINRMS_COOP_STAMPS
#define T16_DBG_DECL uint32_t dbg = 0;
#define T16_DBG_RETURN if (slow_counts) *slow_counts = dbg;
#else
#define T16_DBG_DECL
#define T16_DBG_RETURN
#endif

// Runs all rows; returns the last row's value in column jl (0 if jl is not in the panel).
// A macro, so that the function exists once per panel width: the T16_W_... macros in force where
// it is expanded decide which.
#define T16_DEFINE_TASK_FUNCTION(NAME) \
__device__ __forceinline__ float NAME(const TaskAsmArgs &a, uint32_t *slow_counts = nullptr) \
{ \
  float cv = 0.0f; \
  T16_DBG_DECL \
  uint64_t bp = a.bp_addr, xo = a.xout_addr; \
  const uint64_t crow = t16_uniform64((uint64_t)(uintptr_t)a.crow), bpstr = t16_uniform64(a.bp_stride_bytes); \
  const uint64_t errp = t16_uniform64((uint64_t)(uintptr_t)a.err); \
  const uint32_t cstr = t16_uniform(a.cstride_bytes), nrows = t16_uniform((uint32_t)a.nrows), flags = t16_uniform(a.flags); \
  const uint32_t gbase = t16_uniform((uint32_t)a.gbase), pring = t16_uniform(a.pring_at), prow = t16_uniform(a.pring_row_bytes); \
  const uint32_t spin = t16_uniform((uint32_t)a.spin_limit), jl = t16_uniform((uint32_t)a.jl); \
  switch (flags) { \
    T16_CASE(0, 0, 0, 0) T16_CASE(1, 0, 0, 0) T16_CASE(0, 1, 0, 0) T16_CASE(1, 1, 0, 0) \
    T16_CASE(0, 0, 1, 0) T16_CASE(1, 0, 1, 0) T16_CASE(0, 1, 1, 0) T16_CASE(1, 1, 1, 0) \
    T16_CASE(0, 0, 2, 0) T16_CASE(1, 0, 2, 0) T16_CASE(0, 1, 2, 0) T16_CASE(1, 1, 2, 0) \
    T16_CASE(0, 0, 0, 1) T16_CASE(1, 0, 0, 1) T16_CASE(0, 1, 0, 1) T16_CASE(1, 1, 0, 1) \
    T16_CASE(0, 0, 1, 1) T16_CASE(1, 0, 1, 1) T16_CASE(0, 1, 1, 1) T16_CASE(1, 1, 1, 1) \
    T16_CASE(0, 0, 2, 1) T16_CASE(1, 0, 2, 1) T16_CASE(0, 1, 2, 1) T16_CASE(1, 1, 2, 1) \
    default: break; \
  } \
  T16_DBG_RETURN \
  return cv; \
}

T16_DEFINE_TASK_FUNCTION(dp_task_rows_asm16)

// ---- 32 columns
#undef T16_W_MIDDLE
#undef T16_W_FIRST_ADVANCE
#undef T16_W_INIT_MORE
#undef T16_W_PICKS_MORE
#undef T16_W_LASTCOL
#undef T16_W_INJECT_MORE
#undef T16_W_PUBWRITES_MORE
#undef T16_W_SLOTSHIFT
#undef T16_W_NWRITESPUB_0
#undef T16_W_NWRITESPUB_1
#undef T16_W_NWRITESPUB_2
#define T16_W_MIDDLE(...) T16_MIDDLE_32(__VA_ARGS__)
#define T16_W_FIRST_ADVANCE "s_mov_b32 s93, 0\n\t"         /* 32 columns: the second half of row 0 is loaded in the row */
#define T16_W_INIT_MORE                                                                \
  "v_mov_b32 v80, %[binit]\n\t" "v_mov_b32 v81, %[binit]\n\t" "v_mov_b32 v82, %[binit]\n\t" "v_mov_b32 v83, %[binit]\n\t" \
  "v_mov_b32 v84, %[binit]\n\t" "v_mov_b32 v85, %[binit]\n\t" "v_mov_b32 v86, %[binit]\n\t" "v_mov_b32 v87, %[binit]\n\t" \
  "v_mov_b32 v88, %[binit]\n\t" "v_mov_b32 v89, %[binit]\n\t" "v_mov_b32 v90, %[binit]\n\t" "v_mov_b32 v91, %[binit]\n\t" \
  "v_mov_b32 v92, %[binit]\n\t" "v_mov_b32 v93, %[binit]\n\t" "v_mov_b32 v94, %[binit]\n\t" "v_mov_b32 v95, %[binit]\n\t"
#define T16_W_PICKS_MORE                                                               \
  T16_PICK("16", "v80") T16_PICK("17", "v81") T16_PICK("18", "v82") T16_PICK("19", "v83") \
  T16_PICK("20", "v84") T16_PICK("21", "v85") T16_PICK("22", "v86") T16_PICK("23", "v87") \
  T16_PICK("24", "v88") T16_PICK("25", "v89") T16_PICK("26", "v90") T16_PICK("27", "v91") \
  T16_PICK("28", "v92") T16_PICK("29", "v93") T16_PICK("30", "v94") T16_PICK("31", "v95")
#define T16_W_LASTCOL "v95"
#define T16_W_INJECT_MORE                                                              \
  T16_INJECT_8TO15                                                                     \
  "ds_read2_b32 v[80:81], v103 offset0:17 offset1:18\n\t"                              \
  "ds_read2_b32 v[82:83], v103 offset0:19 offset1:20\n\t"                              \
  "ds_read2_b32 v[84:85], v103 offset0:21 offset1:22\n\t"                              \
  "ds_read2_b32 v[86:87], v103 offset0:23 offset1:24\n\t"                              \
  "ds_read2_b32 v[88:89], v103 offset0:25 offset1:26\n\t"                              \
  "ds_read2_b32 v[90:91], v103 offset0:27 offset1:28\n\t"                              \
  "ds_read2_b32 v[92:93], v103 offset0:29 offset1:30\n\t"                              \
  "ds_read2_b32 v[94:95], v103 offset0:31 offset1:32\n\t"
#define T16_W_PUBWRITES_MORE                                                           \
  T16_PUBWRITES_8TO15                                                                  \
  "ds_write_b128 v102, v[80:83] offset:64\n\t"                                         \
  "ds_write_b128 v102, v[84:87] offset:80\n\t"                                         \
  "ds_write_b128 v102, v[88:91] offset:96\n\t"                                         \
  "ds_write_b128 v102, v[92:95] offset:112\n\t"
#define T16_W_SLOTSHIFT "7"
#define T16_W_NWRITESPUB_0 "9"
#define T16_W_NWRITESPUB_1 "10"
#define T16_W_NWRITESPUB_2 "9"

T16_DEFINE_TASK_FUNCTION(dp_task_rows_asm32)

// ---- 8 columns
#undef T16_W_MIDDLE
#undef T16_W_SETA
#undef T16_W_SETB
#undef T16_W_FIRST_LOAD
#undef T16_W_FIRST_ADVANCE
#undef T16_W_INIT_MORE
#undef T16_W_PICKS_MORE
#undef T16_W_LASTCOL
#undef T16_W_INJECT_MORE
#undef T16_W_PUBWRITES_MORE
#undef T16_W_SLOTSHIFT
#undef T16_W_NWRITESPUB_0
#undef T16_W_NWRITESPUB_1
#undef T16_W_NWRITESPUB_2
#define T16_W_MIDDLE(...) T16_MIDDLE_8(__VA_ARGS__)
#define T16_W_SETA T16_SETA_8
#define T16_W_SETB T16_SETB_8
#define T16_W_FIRST_LOAD "s_load_dwordx8 s[40:47], s[72:73], 0x4\n\t"
#define T16_W_FIRST_ADVANCE "s_mov_b32 s93, %[cstr]\n\t"
#define T16_W_INIT_MORE
#define T16_W_PICKS_MORE
#define T16_W_LASTCOL "v71"
#define T16_W_INJECT_MORE
#define T16_W_PUBWRITES_MORE
#define T16_W_SLOTSHIFT "5"
#define T16_W_NWRITESPUB_0 "3"
#define T16_W_NWRITESPUB_1 "4"
#define T16_W_NWRITESPUB_2 "3"

T16_DEFINE_TASK_FUNCTION(dp_task_rows_asm8)

#undef T16_CASE
#undef T16_DBG_DECL
#undef T16_DBG_RETURN
#undef T16_DEFINE_TASK_FUNCTION
#undef T16_COUNT
#undef T16_TASK
#undef T16_PHASES_0
#undef T16_PHASES_1
#undef T16_SLOWS_0
#undef T16_SLOWS_1
#undef T16_SETA
#undef T16_SETB
#undef T16_ROW_
#undef T16_PICK
#undef T16_ONE
#undef T16_CELLS
#undef T16_POLL
#undef T16_INJECT_READS
#undef T16_ROW
#undef T16_SLOW

}  // namespace dp
