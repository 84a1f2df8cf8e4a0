// orb_geom.h — per-resolution geometry of the extractor, built once on the host (orb.hip) and read by
// every kernel through scalar loads.  All formulas follow the reference lines cited in orb.hip.
#pragma once
#include <stdint.h>
#include "../../include/dvslam_hip.h"

namespace dvs {

constexpr int kEdge = 19;        // EDGE_THRESHOLD (ORBextractor.cpp:73)
constexpr int kMinBorder = 16;   // EDGE_THRESHOLD - 3 (ORBextractor.cpp:789-790)
constexpr int kHalfPatch = 15;   // HALF_PATCH_SIZE (ORBextractor.cpp:72)
constexpr int kMaxCellDim = 76;  // wCell < 70 whenever nCols >= 1, +6 overlap
constexpr int kTilePitch = 80;   // LDS row pitch of a FAST cell tile (bytes)
constexpr int kMaxQuota = 1500;  // per-level keypoint quota supported by the LDS-resident quad-tree

struct LevelGeom {
  int32_t w, h, pitch;       // level image size, bytes between rows in the pyramid / blurred blocks
  int32_t nCols, nRows, wCell, hCell;
  int32_t cellBase, nCells;  // range of this level in the frame's flat cell table (non-skipped cells, row-major)
  int32_t cellCap;           // candidate slots per cell = ceil(wCell/2) * ceil(hCell/2) (strict 3x3 maxima)
  int32_t ptsCap;            // nCells * cellCap
  int32_t N;                 // mnFeaturesPerLevel[level]
  int32_t nIni;              // quad-tree roots
  float hX;                  // root width (float, ORBextractor.cpp:561)
  int32_t regionW, regionH;  // maxBorder - minBorder
  int32_t kpOff;             // first slot of this level in the per-frame level-keypoint block (N + 4 slots)
  int32_t xtab, ytab;        // offsets of this level's resize tables (level >= 1)
  int32_t gtab;              // offset of this level's ResizeGroup table
  float scale;               // mvScaleFactor[level]
  float kpSize;              // (float)(int)(31 * scale)
  uint64_t off;              // byte offset of the level inside a frame's pyramid (and blurred) block
  uint64_t candOff;          // uint32 index of the level's first cell slot inside a frame's candidate block
  uint64_t ptsOff;           // uint32 index of the level's linear point list inside a frame's points block
};

struct Geom {
  int32_t nlevels, rows, cols;
  int32_t totalCells;   // over all levels
  int32_t kpBlock;      // sum over levels of (N + 4)
  int32_t outCap;       // nfeatures + 3 * nlevels
  int32_t blurTiles;    // tiles of the (generic) blur launch over all levels
  int32_t blurStrips;   // wave work items of the streaming blur
  int32_t blurBand;     // rows per work item: kBlurBand for batches, fewer for a few frames (a wavefront walks its rows one after the other)
  int32_t pyrTiles;     // workgroups of the pyramid cascade
  int32_t pyrLds;       // bytes of ONE of its two LDS buffers
  int32_t fastP;        // LDS tile pitch of the wave-per-cell FAST kernel (48 / 64 / 80)
  int32_t fastRows;     // max cell height (rows of the LDS tile; <= fastP + 4)
  int32_t fastWaveLds;  // LDS bytes per wave: tile (fast_tile_bytes(fastP)) + score tile (fastRows * fastP) + work list
  int32_t fastTile;     // = fast_tile_bytes(fastP)
  int32_t fastByteDma;  // 1: the LDS-DMA takes byte-aligned global addresses here (checked at start-up): tiles start 1 column left of the cell
  int32_t iniTh, minTh;
  int32_t maxN;         // max quota over levels
  int32_t gk[7];
  int32_t umax[16];
  // intensity-centroid patch as per-lane byte weights: lane = 2*row + half covers 16 bytes of one patch row;
  // [0..3] = (u + 15) where the pixel is inside the circular patch else 0, [4..7] = 1 / 0 membership
  uint32_t icw[64][8];
  uint64_t frameBytes;  // pyramid block per frame
  uint64_t candPerFrame, ptsPerFrame;  // uint32 elements per frame
  LevelGeom lv[DVS_MAX_LEVELS];
};

// bytes of k_fast_wave<P>'s tile: up to P + 4 rows of P bytes (1280x720: cells of up to 44 x 49 at pitch 48), a multiple of 256
// (whole LDS-DMA wave-instructions)
constexpr int fast_tile_bytes(int P) { return ((P + 4) * P + 255) & ~255; }

struct Cell {       // one FAST cell (ORBextractor.cpp:805-827)
  int16_t level, i, j;
  int16_t rpt;             // k_fast_wave: rows per trip of the rejection loop = 64 / ng (ng = 4-pixel column groups of the interior)
  int16_t x0, y0, cw, ch;  // sub-image origin and size in level pixels (rowRange/colRange)
  int32_t slot;            // index among the level's cells (candidate order)
  float inv_ng;            // 1.0f / ng — host-made: the two divisions were ~12 vector instructions per cell on the device
};

struct BlurTile { int16_t level, tx, ty, pad; };

// resize: one entry per group of 4 output columns — dword-aligned source byte `base`, bit shift that moves the group's first
// tap to byte 0 of an 8-byte window, one v_perm selector per column that places (left tap, right tap) into the two 16-bit
// halves, and the Q11 coefficient pairs (a0 | a1 << 16) for v_dot2_u32_u16
struct ResizeGroup { int32_t base; uint32_t shift; uint32_t sel[4]; int32_t alpha[4]; };

// streaming blur work item: one wavefront filters a strip of `w` columns (4 per lane) x Geom::blurBand rows
constexpr int kBlurBand = 64;
struct BlurStrip { int16_t level, x0, w, y0; };

// matrix-core blur work item: one workgroup filters the 128-column super-strip `strip` of a level (wavefront w its 32-column strip
// 4 strip + w) over `nt` tiles of 32 rows from tile t0; tab = index of the first strip's pair of horizontal operand fragments
constexpr int kBlurMfmaBand = 8;
struct BlurCol { int16_t level, strip, t0, nt; int32_t tab; };

// pyramid cascade: one workgroup builds its share of EVERY level from one staged level-0 region (LDS ping-pong).
// Ownership: level-1 tiles partition level 1; a pixel of level k >= 2 belongs to the tile that owns its top-left source tap,
// so each level is partitioned too.  c* = region that must be computed (owned + what deeper levels read), o* = owned part.
struct PyrTileLevel { int16_t cx0, cx1, cy0, cy1, ox0, ox1, oy0, oy1; };
struct PyrTile { int16_t sx0, sx1, sy0, sy1; PyrTileLevel lv[DVS_MAX_LEVELS]; };
constexpr int kPyrTileW = 128, kPyrTileH = 64;

// packed candidate / keypoint: x (12 bits) | y (12 bits) << 12 | score << 24, region-relative coordinates
__host__ __device__ inline uint32_t pack_pt(int x, int y, int s) { return (uint32_t)x | ((uint32_t)y << 12) | ((uint32_t)s << 24); }
__host__ __device__ inline int pt_x(uint32_t p) { return (int)(p & 0xFFFu); }
__host__ __device__ inline int pt_y(uint32_t p) { return (int)((p >> 12) & 0xFFFu); }
__host__ __device__ inline int pt_s(uint32_t p) { return (int)(p >> 24); }

}  // namespace dvs
