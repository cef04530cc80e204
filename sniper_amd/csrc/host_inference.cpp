// host_inference.cpp -- host-only code of the library (no kernels: a .cpp, so that the identity of the DEVICE sources that
// bench.py keys its counter reports on -- every .hip / .h of this directory -- does not move with it).
#include "common.h"
#include <math.h>
#include <string.h>
#include <vector>

// ---------------------------------------------------------------------------------------------
// focus_chips_host: `gmask` of lib/chips/chips_inference.py:12-89 on the HOST, in one call -- threshold the FocusPixel map, dilate
// (cv2.dilate with a d x d kernel), bounding rectangles of the contours cv2.findContours(RETR_LIST) reports (every 8-connected
// foreground component, and every hole = 4-connected background component that does not touch the map border, grown by one
// cell), minimum side `ms`, paint-and-repeat until the chip count is stable, x16, clamp to the crop, / cscale.  The maps are at
// most 125 x 88 cells: this is host work in the reference too (OpenCV); the Python statement of the same steps over
// scipy.ndimage (sniper_amd/chips_inference.py::gmask_reference, 250 us per map) stays as the definition this is tested against,
// component order included (raster order of a component's first cell, as scipy.ndimage.label numbers them).
// The cv2 calls themselves are restated, not linked: PARITY UNPINNED, exactly as for the Python statement.
namespace {
struct CellRect { int x, y, w, h; };

// The map lives in a frame of one sentinel cell (value 1: neither background 0 nor foreground 255), so a flood fill needs no
// bounds checks.  Bounding boxes of the connected components of `want`-valued cells (8- or 4-connectivity), in raster order of a
// component's first cell; holes_only: components that touch the map border are skipped and the box grows by one cell.
void component_rects(const unsigned char *mp, int H, int W, unsigned char want, bool conn8, bool holes_only, unsigned char *seen,
                     int *stack, CellRect *out, int cap, int *n, int wy0, int wy1, int wx0, int wx1) {
  const int Wp = W + 2;
  const int off8[8] = {-Wp - 1, -Wp, -Wp + 1, -1, 1, Wp - 1, Wp, Wp + 1}, off4[4] = {-Wp, -1, 1, Wp};
  const int dx8[8] = {-1, 0, 1, -1, 1, -1, 0, 1}, dx4[4] = {0, -1, 1, 0};
  const int *off = conn8 ? off8 : off4, *dxs = conn8 ? dx8 : dx4;
  const int noff = conn8 ? 8 : 4;
  // only the window [wy0, wy1] x [wx0, wx1] (map coordinates) is walked: everything outside it counts as visited
  memset(seen, 1, (size_t)(H + 2) * Wp);
  for (int y = wy0; y <= wy1; ++y) memset(seen + (y + 1) * Wp + wx0 + 1, 0, (size_t)(wx1 - wx0 + 1));
  for (int y = wy0 + 1; y <= wy1 + 1; ++y)
    for (int x = wx0 + 1; x <= wx1 + 1; ++x) {
      const int i0 = y * Wp + x;
      if (mp[i0] != want || seen[i0]) continue;
      // the stack holds (cell, its column): rows follow from the smallest / largest cell index, no division per cell
      int x0 = x, x1 = x, lo = i0, hi = i0, top = 0;
      stack[top++] = i0;
      stack[top++] = x;
      seen[i0] = 1;
      while (top) {
        const int cx = stack[--top], i = stack[--top];
        x0 = cx < x0 ? cx : x0; x1 = cx > x1 ? cx : x1; lo = i < lo ? i : lo; hi = i > hi ? i : hi;
        for (int k = 0; k < noff; ++k) {
          const int j = i + off[k];
          if (mp[j] == want && !seen[j]) {
            seen[j] = 1;
            stack[top++] = j;
            stack[top++] = cx + dxs[k];
          }
        }
      }
      int y0 = lo / Wp, y1 = hi / Wp;
      --x0; --x1; --y0; --y1;                      // frame coordinates -> map coordinates
      if (holes_only) {
        // open background, not a hole: it reaches a side of the window, and every side of the window is either the map border
        // or a foreground-free ring around the foreground's bounding box (connected to the map border outside it)
        if (x0 == wx0 || y0 == wy0 || x1 == wx1 || y1 == wy1) continue;
        if (*n < cap) out[*n] = CellRect{x0 - 1, y0 - 1, x1 - x0 + 3, y1 - y0 + 3};
      } else if (*n < cap) {
        out[*n] = CellRect{x0, y0, x1 - x0 + 1, y1 - y0 + 1};
      }
      ++*n;
    }
}

// The same rectangles in the ORDER cv2.findContours(RETR_LIST) reports them (round 5): border following of Suzuki & Abe (CVGIP 30,
// 1985, Algorithm 1; OpenCV's contours.cpp) on the framed map -- an outer border starts at a foreground cell whose left neighbour is
// background, a hole border at a (not yet right-edge-marked) foreground cell whose right neighbour is background; borders are
// followed with the 8-neighbourhood, cells marked NBD / -NBD; every border's bounding rectangle is recorded and the list is returned
// newest first (OpenCV links each finished contour in front of the earlier ones).  f: int scratch of (H + 2) * (W + 2).
// Restated and tested against the component form above and against oracle/cv_contours.py (rectangles AND order).
void border_rects(const unsigned char *mp, int H, int W, int *f, CellRect *out, int cap, int *n) {
  const int Wp = W + 2;
  static const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1}, DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};     // counter-clockwise on the screen from east
  for (int y = 0; y < H + 2; ++y)
    for (int x = 0; x < Wp; ++x) f[y * Wp + x] = (y >= 1 && y <= H && x >= 1 && x <= W && mp[y * Wp + x] == 255) ? 1 : 0;
  *n = 0;
  int nbd = 1;
  for (int i = 1; i <= H; ++i)
    for (int j = 1; j <= W; ++j) {
      const int v = f[i * Wp + j];
      if (v == 0) continue;
      int k0;
      if (v == 1 && f[i * Wp + j - 1] == 0) k0 = 4;             // outer border: start the clockwise search at the west neighbour
      else if (v >= 1 && f[i * Wp + j + 1] == 0) k0 = 0;        // hole border: at the east neighbour
      else continue;
      ++nbd;
      int x0 = j, x1 = j, y0 = i, y1 = i;
      int i1 = -1, j1 = -1;
      for (int s = 0; s < 8; ++s) {
        const int k = (k0 - s + 8) & 7;
        if (f[(i + DY[k]) * Wp + j + DX[k]] != 0) { i1 = i + DY[k]; j1 = j + DX[k]; break; }
      }
      if (i1 < 0) {
        f[i * Wp + j] = -nbd;
      } else {
        int i2 = i1, j2 = j1, i3 = i, j3 = j;
        for (;;) {
          int k = 0;
          for (; k < 8; ++k)
            if (DY[k] == i2 - i3 && DX[k] == j2 - j3) break;
          bool east_zero = false;
          int i4 = i3, j4 = j3;
          for (int s = 1; s <= 8; ++s) {
            const int kk = (k + s) & 7;
            if (f[(i3 + DY[kk]) * Wp + j3 + DX[kk]] != 0) { i4 = i3 + DY[kk]; j4 = j3 + DX[kk]; break; }
            if (kk == 0) east_zero = true;
          }
          if (east_zero) f[i3 * Wp + j3] = -nbd;
          else if (f[i3 * Wp + j3] == 1) f[i3 * Wp + j3] = nbd;
          if (i4 == i && j4 == j && i3 == i1 && j3 == j1) break;
          i2 = i3; j2 = j3; i3 = i4; j3 = j4;
          x0 = j3 < x0 ? j3 : x0; x1 = j3 > x1 ? j3 : x1; y0 = i3 < y0 ? i3 : y0; y1 = i3 > y1 ? i3 : y1;
        }
      }
      if (*n < cap) out[*n] = CellRect{x0 - 1, y0 - 1, x1 - x0 + 1, y1 - y0 + 1};
      ++*n;
    }
  const int m = *n < cap ? *n : cap;
  for (int a = 0, b = m - 1; a < b; ++a, --b) { const CellRect t = out[a]; out[a] = out[b]; out[b] = t; }      // newest first
}

CellRect place_rect(CellRect r, int ms, int iw, int ih) {
  const int cx = (r.x + r.x + r.w) / 2, cy = (r.y + r.y + r.h) / 2;
  const int w = r.w > ms ? r.w : ms, h = r.h > ms ? r.h : ms;
  int x, y;
  if (cx + w / 2 >= iw) x = iw - w >= 0 ? iw - w : 0;
  else if (cx - w / 2 < 0) x = 0;
  else x = cx - w / 2;
  if (cy + h / 2 >= ih) y = ih - h >= 0 ? ih - h : 0;
  else if (cy - h / 2 < 0) y = 0;
  else y = cy - h / 2;
  return CellRect{x, y, w, h};
}
}  // namespace

// Bounding rectangles of the contours of a 0 / 255 mask, as cv2.findContours(RETR_LIST) + cv2.boundingRect give them.
// mode 0: border following, cv2's order (what sn_focus_chips_host uses); mode 1: connected components + enclosed holes, raster order
// of their first cells (the round 2-4 form; kept as the independent cross-check).  rects_xywh (cap, 4) int32.
SN_EXPORT int sn_focus_rects_host(const uint8_t *mask_hw, int H, int W, int mode, int32_t *rects_xywh, int cap, int32_t *n_rects) {
  SN_REQUIRE(mask_hw && rects_xywh && n_rects && H > 0 && W > 0 && H * W <= (1 << 20) && cap > 0, "sn_focus_rects_host: bad arguments");
  const int Wp = W + 2;
  const size_t cells = (size_t)(H + 2) * Wp;
  std::vector<unsigned char> m(cells, 1), seen(cells);
  std::vector<int> stack(cells * 2);
  std::vector<CellRect> rects((size_t)cap);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) m[(y + 1) * Wp + x + 1] = mask_hw[y * W + x] ? 255 : 0;
  int n = 0;
  if (mode == 0) {
    border_rects(m.data(), H, W, stack.data(), rects.data(), cap, &n);
  } else {
    component_rects(m.data(), H, W, 255, true, false, seen.data(), stack.data(), rects.data(), cap, &n, 0, H - 1, 0, W - 1);
    if (n > 0 && n <= cap)
      component_rects(m.data(), H, W, 0, false, true, seen.data(), stack.data(), rects.data(), cap, &n, 0, H - 1, 0, W - 1);
  }
  SN_REQUIRE(n <= cap, "sn_focus_rects_host: more contours than the caller's capacity");
  for (int k = 0; k < n; ++k) {
    rects_xywh[4 * k] = rects[k].x; rects_xywh[4 * k + 1] = rects[k].y; rects_xywh[4 * k + 2] = rects[k].w; rects_xywh[4 * k + 3] = rects[k].h;
  }
  *n_rects = n;
  return SN_OK;
}

SN_EXPORT int sn_focus_chips_host(const float *map_hw, int H, int W, int d, float thresh, int ms, double im_width, double im_height,
                                  double cscale, double *chips_xyxy, int max_chips, int32_t *n_chips) {
  SN_REQUIRE(map_hw && chips_xyxy && n_chips && H > 0 && W > 0 && H * W <= (1 << 20) && max_chips > 0 && cscale > 0.0,
             "sn_focus_chips_host: bad arguments");
  const int iw = (int)ceil(im_width / 16.0), ih = (int)ceil(im_height / 16.0);
  const int cap = 4096, Wp = W + 2;
  const size_t cells = (size_t)(H + 2) * Wp;
  // per-thread scratch that only grows: a 128 KB malloc per call is an mmap / munmap pair (page faults: most of the call's time)
  thread_local std::vector<unsigned char> bytes;
  thread_local std::vector<int> ints;
  thread_local std::vector<CellRect> boxes;
  if (bytes.size() < cells * 3) bytes.resize(cells * 3);
  if (ints.size() < cells * 2) ints.resize(cells * 2);
  if (boxes.size() < (size_t)cap * 2) boxes.resize((size_t)cap * 2);
  unsigned char *m = bytes.data();                            // framed map, threshold scratch, flood-fill marks
  int *stack = ints.data();
  CellRect *rects = boxes.data();
  unsigned char *t = m + cells, *seen = m + 2 * cells;
  CellRect *chips = rects + cap;
  memset(m, 1, cells);                                        // the frame (the interior is written below)
  memset(t, 0, cells);
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) t[(y + 1) * Wp + x + 1] = map_hw[y * W + x] >= thresh ? 1 : 0;
  // cv2.dilate(d x d): anchor (d / 2, d / 2) -> window [y - d/2, y + d - 1 - d/2], cells outside the map do not contribute
  const int lo = d > 1 ? d / 2 : 0, hi = d > 1 ? d - 1 - d / 2 : 0;
  // (output cell y sees source rows [y - lo, y + hi]: source row yy reaches output rows [yy - hi, yy + lo]; the few set cells paint)
  for (int y = 0; y < H; ++y) memset(m + (y + 1) * Wp + 1, 0, (size_t)W);
  for (int yy = 0; yy < H; ++yy)
    for (int xx = 0; xx < W; ++xx) {
      if (!t[(yy + 1) * Wp + xx + 1]) continue;
      for (int y = (yy - hi < 0 ? 0 : yy - hi); y <= yy + lo && y < H; ++y)
        for (int x = (xx - hi < 0 ? 0 : xx - hi); x <= xx + lo && x < W; ++x) m[(y + 1) * Wp + x + 1] = 255;
    }
  // cv2's own procedure and order: border following (border_rects); sn_focus_rects_host(mode 1) keeps the component form of rounds
  // 2 - 4 reachable as the cross-check (same rectangles, raster order)
  auto all_rects = [&](int *n) { border_rects(m, H, W, stack, rects, cap, n); };
  int nr = 0, nchips = -1, nc = 0;
  all_rects(&nr);
  bool overflow = nr > cap;
  while (!overflow && nchips != nc) {
    nchips = nc;
    nc = 0;
    bool painted = false;
    for (int k = 0; k < nr; ++k) {
      const CellRect r = place_rect(rects[k], ms, iw, ih);
      for (int y = r.y; y < r.y + r.h && y < H; ++y)          // numpy slicing clips at the map's edge
        for (int x = r.x; x < r.x + r.w && x < W; ++x) {
          unsigned char &c = m[(y + 1) * Wp + x + 1];
          painted |= c != 255;
          c = 255;
        }
    }
    if (painted) all_rects(&nr);                              // (an unchanged map has the contours it had)
    if (nr > cap) { overflow = true; break; }
    for (int k = 0; k < nr; ++k) chips[nc++] = place_rect(rects[k], ms, iw, ih);
  }
  int status = SN_OK;
  if (overflow || nc > max_chips) {
    status = 1;
  } else {
    for (int k = 0; k < nc; ++k) {
      double x1 = chips[k].x * 16.0, y1 = chips[k].y * 16.0, x2 = (chips[k].x + chips[k].w) * 16.0, y2 = (chips[k].y + chips[k].h) * 16.0;
      if (x2 > im_width) {
        x2 = im_width;
        x1 = fmax(fmin(x1, x2 - ms * 16.0), 0.0);
      }
      if (y2 > im_height) {
        y2 = im_height;
        y1 = fmax(fmin(y1, y2 - ms * 16.0), 0.0);
      }
      chips_xyxy[4 * k + 0] = x1 / cscale; chips_xyxy[4 * k + 1] = y1 / cscale;
      chips_xyxy[4 * k + 2] = x2 / cscale; chips_xyxy[4 * k + 3] = y2 / cscale;
    }
    *n_chips = nc;
  }
  SN_REQUIRE(status == SN_OK, "sn_focus_chips_host: more contours / chips than the caller's capacity");
  return SN_OK;
}


// ---------------------------------------------------------------------------------------------
// aggregate_problems_host: the regrouping half of Tester.aggregate (lib/inference.py:166-190).  The reference walks
// classes x images x scales x chips in Python, filters every chip's rows by the scale's valid range (areas in float32,
// _valid_range_filter :176-186) and stacks what is left per (image, class) for the NMS pool.  Here the chips arrive as the GPU
// returned them (rows of one chip grouped by class, float64, + rows per class: sn_det_compact) and ONE pass writes the float32
// rows of all (image, class) problems back to back, in (image, class, scale, chip, row) order, plus the rows per problem --
// exactly the array the batched soft-NMS launch uploads.  Tested equal to the numpy statement (sniper_amd/inference.py).
// part_rows[p]: float64 (n_p, 5) rows of part p; lens (P, nc): rows per class of part p; parts of image i are
// part_of_image[i] .. part_of_image[i + 1] - 1 in (scale, chip) order; range2 (P, 2): squared valid range as float32, <= 0: none.
SN_EXPORT int sn_aggregate_problems_host(const uint64_t *part_rows, const int64_t *lens, const int32_t *part_of_image,
                                         const float *range2, int P, int nc, int num_images, float *out_rows, long capacity_rows,
                                         int64_t *out_sizes, int64_t *total_rows) {
  SN_REQUIRE(part_rows && lens && part_of_image && range2 && out_rows && out_sizes && total_rows && P >= 0 && nc > 0 &&
                 num_images >= 0 && capacity_rows >= 0,
             "sn_aggregate_problems_host: bad arguments");
  thread_local std::vector<int64_t> first;           // first row of class j inside part p
  first.resize((size_t)P * nc);
  for (int p = 0; p < P; ++p) {
    int64_t at = 0;
    for (int j = 0; j < nc; ++j) {
      first[(size_t)p * nc + j] = at;
      at += lens[(size_t)p * nc + j];
    }
  }
  long n = 0;
  for (int i = 0; i < num_images; ++i)
    for (int j = 0; j < nc; ++j) {
      const long before = n;
      for (int p = part_of_image[i]; p < part_of_image[i + 1]; ++p) {
        const double *src = reinterpret_cast<const double *>(part_rows[p]) + first[(size_t)p * nc + j] * 5;
        const int64_t cnt = lens[(size_t)p * nc + j];
        const float lo2 = range2[2 * p], hi2 = range2[2 * p + 1];
        for (int64_t r = 0; r < cnt; ++r, src += 5) {
          const float x1 = (float)src[0], y1 = (float)src[1], x2 = (float)src[2], y2 = (float)src[3];
          const float area = (y2 - y1) * (x2 - x1);                  // float32 products, as _valid_range_filter
          if (lo2 > 0.f && !(area > lo2)) continue;
          if (hi2 > 0.f && !(area <= hi2)) continue;
          SN_REQUIRE(n < capacity_rows, "sn_aggregate_problems_host: output capacity");
          float *dst = out_rows + n * 5;
          dst[0] = x1; dst[1] = y1; dst[2] = x2; dst[3] = y2; dst[4] = (float)src[4];
          ++n;
        }
      }
      out_sizes[(size_t)i * nc + j] = n - before;
    }
  *total_rows = n;
  return SN_OK;
}
