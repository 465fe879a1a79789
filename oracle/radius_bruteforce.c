/* ORACLE (test infrastructure only) -- plain-C restatement of the predicate
 * behind the reference's radius graph (models/graph_gen.py:207-209):
 * scikit-learn's ball-tree radius query on float64 data returns point p for
 * centre c iff  ((px-cx)^2 + (py-cy)^2) + (pz-cz)^2 <= r*r  in float64
 * (inclusive).  O(P*Q), no tree, no early exit.  Built with
 * -ffp-contract=off so the sum is three separately rounded products.
 * Output rows are (point_idx, centre_idx), ascending centre then point.
 * Never linked into the product library. */
#include <stdint.h>

long long oracle_radius_count(const double *p, long long np_, const double *c,
                              long long nc, double r, int64_t *counts) {
  const double r2 = r * r;
  long long total = 0;
  for (long long q = 0; q < nc; ++q) {
    const double cx = c[3 * q], cy = c[3 * q + 1], cz = c[3 * q + 2];
    int64_t n = 0;
    for (long long i = 0; i < np_; ++i) {
      const double dx = p[3 * i] - cx, dy = p[3 * i + 1] - cy,
                   dz = p[3 * i + 2] - cz;
      const double d2 = (dx * dx + dy * dy) + dz * dz;
      n += (d2 <= r2);
    }
    counts[q] = n;
    total += n;
  }
  return total;
}

void oracle_radius_fill(const double *p, long long np_, const double *c,
                        long long nc, double r, int64_t *edges) {
  const double r2 = r * r;
  long long e = 0;
  for (long long q = 0; q < nc; ++q) {
    const double cx = c[3 * q], cy = c[3 * q + 1], cz = c[3 * q + 2];
    for (long long i = 0; i < np_; ++i) {
      const double dx = p[3 * i] - cx, dy = p[3 * i + 1] - cy,
                   dz = p[3 * i + 2] - cz;
      const double d2 = (dx * dx + dy * dy) + dz * dz;
      if (d2 <= r2) {
        edges[2 * e] = i;
        edges[2 * e + 1] = q;
        ++e;
      }
    }
  }
}
