/* oracle/lap.c — C restatement of the rectangular linear-sum-assignment solver behind
 * scipy.optimize.linear_sum_assignment (TEST INFRASTRUCTURE, see oracle/__init__.py).
 *
 * The reference calls scipy from adapteacher/modeling/GModule/utils/hungarian.py:63
 * (scipy==1.7.3 pinned in requirements.txt:83; un-vendored third-party dependency, 1.15.3
 * installed in this image).  scipy's solver is the shortest-augmenting-path algorithm of
 * D. F. Crouse, "On implementing 2D rectangular assignment algorithms", IEEE TAES 52(4), 2016,
 * restated here from its published description: rows are inserted one at a time, a Dijkstra-like
 * scan over the not-yet-scanned columns finds the shortest augmenting path, the duals u,v are
 * updated and the path is flipped.  What matters for parity are the tie rules, which this file
 * reproduces and tests/test_oracle_lap.py pins against the installed scipy:
 *   - the unscanned-column list starts in REVERSE order (nc-1 .. 0) and removal swaps in the last entry;
 *   - among equal path costs the scan prefers a column that is still unassigned, later positions
 *     overriding earlier ones; otherwise the first minimum in list order wins;
 *   - tall matrices (nr > nc) are transposed first; costs are evaluated in double precision as
 *     ((minVal + c_ij) - u_i) - v_j.
 * The device kernel (ttdg-mgm_amd/csrc/lap.hip) runs the same steps with one wavefront per matrix.
 *
 * int ttdg_oracle_lap(const double* cost, int nr, int nc, int maximize, int64_t* col_of_row)
 *   cost row-major nr x nc; writes for every row i of the ORIGINAL matrix the assigned column or -1.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static int solve_wide(const double* c, int nr, int nc, int* col4row, int* row4col) {
  double* u = calloc(nr, sizeof(double));
  double* v = calloc(nc, sizeof(double));
  double* spc = malloc(nc * sizeof(double));
  int* path = malloc(nc * sizeof(int));
  int* remaining = malloc(nc * sizeof(int));
  char* SR = malloc(nr);
  char* SC = malloc(nc);
  int ok = 1;
  for (int i = 0; i < nr; ++i) col4row[i] = -1;
  for (int j = 0; j < nc; ++j) { row4col[j] = -1; path[j] = -1; }

  for (int cur = 0; cur < nr && ok; ++cur) {
    double minVal = 0;
    int nrem = nc, i = cur, sink = -1;
    for (int it = 0; it < nc; ++it) remaining[it] = nc - it - 1;
    memset(SR, 0, nr);
    memset(SC, 0, nc);
    for (int j = 0; j < nc; ++j) spc[j] = INFINITY;
    while (sink == -1) {
      int index = -1;
      double lowest = INFINITY;
      SR[i] = 1;
      for (int it = 0; it < nrem; ++it) {
        int j = remaining[it];
        double r = minVal + c[(size_t)i * nc + j] - u[i] - v[j];
        if (r < spc[j]) { path[j] = i; spc[j] = r; }
        if (spc[j] < lowest || (spc[j] == lowest && row4col[j] == -1)) { lowest = spc[j]; index = it; }
      }
      minVal = lowest;
      if (minVal == INFINITY) { ok = 0; break; }
      int j = remaining[index];
      if (row4col[j] == -1) sink = j; else i = row4col[j];
      SC[j] = 1;
      remaining[index] = remaining[--nrem];
    }
    if (!ok) break;
    u[cur] += minVal;
    for (int i2 = 0; i2 < nr; ++i2)
      if (SR[i2] && i2 != cur) u[i2] += minVal - spc[col4row[i2]];
    for (int j = 0; j < nc; ++j)
      if (SC[j]) v[j] -= minVal - spc[j];
    int j = sink;
    for (;;) {
      int i2 = path[j];
      row4col[j] = i2;
      int t = col4row[i2]; col4row[i2] = j; j = t;
      if (i2 == cur) break;
    }
  }
  free(u); free(v); free(spc); free(path); free(remaining); free(SR); free(SC);
  return ok ? 0 : -1;
}

int ttdg_oracle_lap(const double* cost, int nr, int nc, int maximize, int64_t* col_of_row) {
  if (nr == 0 || nc == 0) return 0;
  int transpose = nc < nr;
  int R = transpose ? nc : nr, C = transpose ? nr : nc;
  double* t = malloc((size_t)R * C * sizeof(double));
  for (int i = 0; i < nr; ++i)
    for (int j = 0; j < nc; ++j) {
      double x = maximize ? -cost[(size_t)i * nc + j] : cost[(size_t)i * nc + j];
      if (transpose) t[(size_t)j * C + i] = x; else t[(size_t)i * C + j] = x;
    }
  int* c4r = malloc(R * sizeof(int));
  int* r4c = malloc(C * sizeof(int));
  int rc = solve_wide(t, R, C, c4r, r4c);
  if (rc == 0) {
    for (int i = 0; i < nr; ++i) col_of_row[i] = -1;
    if (transpose) { for (int p = 0; p < R; ++p) col_of_row[c4r[p]] = p; }
    else { for (int p = 0; p < R; ++p) col_of_row[p] = c4r[p]; }
  }
  free(t); free(c4r); free(r4c);
  return rc;
}
