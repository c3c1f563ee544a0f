/* agrifly_oracle_render.c -- see agrifly_oracle_render.h (TEST INFRASTRUCTURE ONLY). */
#include "agrifly_oracle_render.h"

#include <math.h>

void ora_quat_mul(const double a[4], const double b[4], double out[4]) {
  /* Rotation.hpp:124-131: (this = a) * (r1 = b) */
  const double c0 = b[0] * a[0] - b[1] * a[1] - b[2] * a[2] - b[3] * a[3];
  const double c1 = b[1] * a[0] + b[0] * a[1] + b[3] * a[2] - b[2] * a[3];
  const double c2 = b[2] * a[0] - b[3] * a[1] + b[0] * a[2] + b[1] * a[3];
  const double c3 = b[3] * a[0] + b[2] * a[1] - b[1] * a[2] + b[0] * a[3];
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void ora_quat_to_matrix(const double q[4], double R[9]) {
  /* Rotation.hpp:196-220 */
  const double r0 = q[0] * q[0], r1 = q[1] * q[1], r2 = q[2] * q[2], r3 = q[3] * q[3];
  R[0] = r0 + r1 - r2 - r3;
  R[1] = 2 * q[1] * q[2] - 2 * q[0] * q[3];
  R[2] = 2 * q[1] * q[3] + 2 * q[0] * q[2];
  R[3] = 2 * q[1] * q[2] + 2 * q[0] * q[3];
  R[4] = r0 - r1 + r2 - r3;
  R[5] = 2 * q[2] * q[3] - 2 * q[0] * q[1];
  R[6] = 2 * q[1] * q[3] - 2 * q[0] * q[2];
  R[7] = 2 * q[2] * q[3] + 2 * q[0] * q[1];
  R[8] = r0 - r1 - r2 + r3;
}

static double dot3(const double a[3], const double b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross3(const double a[3], const double b[3], double o[3]) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}

/* ray o + t d against one triangle; returns t or +inf.  Two-sided, t > 0. */
static double ray_triangle(const double o[3], const double d[3], const float *tri) {
  const double v0[3] = {tri[0], tri[1], tri[2]};
  const double e1[3] = {(double)tri[3] - v0[0], (double)tri[4] - v0[1], (double)tri[5] - v0[2]};
  const double e2[3] = {(double)tri[6] - v0[0], (double)tri[7] - v0[1], (double)tri[8] - v0[2]};
  double p[3], q[3];
  cross3(d, e2, p);
  const double det = dot3(e1, p);
  if (fabs(det) < 1e-12) return INFINITY;
  const double inv = 1.0 / det;
  const double tv[3] = {o[0] - v0[0], o[1] - v0[1], o[2] - v0[2]};
  const double u = dot3(tv, p) * inv;
  if (u < 0.0 || u > 1.0) return INFINITY;
  cross3(tv, e1, q);
  const double v = dot3(d, q) * inv;
  if (v < 0.0 || u + v > 1.0) return INFINITY;
  const double t = dot3(e2, q) * inv;
  return t > 0.0 ? t : INFINITY;
}

static void camera_frame(const double att[4], const double mount[4], double R[9]) {
  double q[4];
  ora_quat_mul(att, mount, q);
  ora_quat_to_matrix(q, R);
}

static double pixel_depth(const ora_camera *cam, const float *triangles, int64_t n_tri, const double o[3],
                          const double R[9], int px, int py) {
  const double u = (px - cam->cx) / cam->focal_length;
  const double v = (py - cam->cy) / cam->focal_length;
  /* camera-frame ray (u, v, 1): t along it IS the camera-frame z of the hit */
  const double d[3] = {R[0] * u + R[1] * v + R[2], R[3] * u + R[4] * v + R[5], R[6] * u + R[7] * v + R[8]};
  double best = INFINITY;
  for (int64_t k = 0; k < n_tri; k++) {
    const double t = ray_triangle(o, d, triangles + 9 * k);
    if (t < best) best = t;
  }
  return best;
}

static uint16_t quantise(const ora_camera *cam, double z) {
  if (!(z < INFINITY)) return (uint16_t)cam->max_count;
  const double c = floor(z / cam->depth_scale);
  return (uint16_t)(c < cam->max_count ? c : cam->max_count);
}

void ora_render_depth(const ora_camera *cam, const float *triangles, int64_t n_tri, const double cam_pos[3],
                      const double att[4], const double mount[4], uint16_t *out) {
  double R[9];
  camera_frame(att, mount, R);
  /* rows are independent: the same per-pixel arithmetic whatever the thread count */
#pragma omp parallel for schedule(dynamic, 4)
  for (int py = 0; py < cam->height; py++)
    for (int px = 0; px < cam->width; px++)
      out[(int64_t)py * cam->width + px] = quantise(cam, pixel_depth(cam, triangles, n_tri, cam_pos, R, px, py));
}

double ora_render_pixel_depth(const ora_camera *cam, const float *triangles, int64_t n_tri,
                              const double cam_pos[3], const double att[4], const double mount[4], int px,
                              int py) {
  double R[9];
  camera_frame(att, mount, R);
  return pixel_depth(cam, triangles, n_tri, cam_pos, R, px, py);
}
