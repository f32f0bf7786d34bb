// mgx_internal.h -- shared between the translation units of libmgx.so (not installed).
#ifndef MGX_INTERNAL_H
#define MGX_INTERNAL_H

#include <stdint.h>

// Records a thread-local error message and returns `status` (so callers can `return mgx_fail(...)`).
int mgx_fail(int status, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

// init_genrand(19650218): the seed-independent first pass of MT19937 init_by_array (levelgen.cpp).
void mgx_mt_init_table(uint32_t out[624]);

// ---- internal cell code (1 byte per cell, x-major like Grid.encode()):
//   bits 3:0  kind    0 unseen*, 1 empty, 2 wall, 3 floor, 4 door-open, 5 key, 6 ball, 7 box, 8 goal,
//                     9 lava, 10 agent*, 11 door-closed, 12 door-locked      (* never stored in a grid)
//   bits 6:4  color   COLOR_TO_IDX (0..6); for kind 10 (full-obs agent marker) the agent direction
//   bit  7    aux     Goal.overlap (terminal goal)
#define MGX_K_EMPTY 1
#define MGX_K_WALL 2
#define MGX_K_FLOOR 3
#define MGX_K_DOOR_OPEN 4
#define MGX_K_KEY 5
#define MGX_K_BALL 6
#define MGX_K_BOX 7
#define MGX_K_GOAL 8
#define MGX_K_LAVA 9
#define MGX_K_AGENT 10
#define MGX_K_DOOR_CLOSED 11
#define MGX_K_DOOR_LOCKED 12
#define MGX_CODE_EMPTY 0x01
// Environment knobs (DESIGN.md has the table).  Form selection -- MGX_PARTIAL_KERNEL, MGX_FULL_KERNEL, MGX_ROLLOUT, MGX_SEED_FORM: which of two
// equivalent kernel forms a handle takes; the tests run both sides of every size rule with them -- is read with getenv in every build.  The
// TUNING knobs (launch shaping, waves per block, level-generator lanes / spans) exist only in builds with -DMGX_TUNING (tools/build_variant.sh):
// the default library does not look at them.
#ifdef MGX_TUNING
#define MGX_TUNE_ENV(name) getenv(name)
#else
#define MGX_TUNE_ENV(name) ((const char *)nullptr)
#endif
#define MGX_CODE_WALL_GREY 0x52 /* Wall() default colour grey=5: what Grid.slice pads with (minigrid.py:469) */

#endif
