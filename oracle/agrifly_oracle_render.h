/*
 * agrifly_oracle_render.h -- CPU checker for the depth-camera renderer
 * (SURVEY.md 8f row f4).
 *
 * TEST INFRASTRUCTURE ONLY (same rules as agrifly_oracle.h).
 *
 * PARITY STATUS: unpinned, and necessarily so -- the reference does not contain
 * a renderer.  Its depth image comes out of AirSim/Unity over RPC
 * (Simulator/Rappids_Simulator/main.cpp:332-336, ImageType::DepthVis) and the
 * orchard scene is not in the tree.  What the
 * reference does fix is the image CONTRACT its planner consumes, and that is
 * what this file restates:
 *   - 8-bit depth counts widened to uint16 (main.cpp:352-354),
 *   - count * depthScale = z-depth in metres, depthScale = far / 256, far = 10 m
 *     (main.cpp:120-122, DepthImagePlanner.cpp:78-84 treat the value as the
 *     camera-frame Z of the pixel),
 *   - pinhole model, focal length = width / 2, principal point = image centre,
 *     pixel (x, y) <-> ray ((x - cx)/f, (y - cy)/f, 1) (main.cpp:360,484-486;
 *     DepthImagePlanner.hpp DeprojectPixelToPoint),
 *   - camera-to-world rotation = vehicle attitude * depthCamAtt, camera origin =
 *     vehicle position (main.cpp:123-125,520-523).
 * The geometry is a brute-force ray / triangle test over every triangle
 * (Moeller-Trumbore, two-sided), the independent check for the engine's BVH
 * traversal.
 */
#ifndef AGRIFLY_ORACLE_RENDER_H
#define AGRIFLY_ORACLE_RENDER_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_camera {
  int width, height;
  double focal_length, cx, cy;
  double depth_scale; /* metres per count */
  int max_count;      /* 255 for the 8-bit DepthVis image */
} ora_camera;

/* q = a * b with the product convention of Common/Common/Math/Rotation.hpp:124-131 */
void ora_quat_mul(const double a[4], const double b[4], double out[4]);
/* row-major rotation matrix of a unit quaternion (Rotation.hpp:196-220) */
void ora_quat_to_matrix(const double q[4], double R[9]);

/* One depth image.  triangles: n_tri x 9 floats (v0 v1 v2, world frame).
 * cam_pos: world position of the camera; att, mount: the camera-to-world
 * rotation is att * mount.  out: height x width counts, row major. */
void ora_render_depth(const ora_camera *cam, const float *triangles, int64_t n_tri, const double cam_pos[3],
                      const double att[4], const double mount[4], uint16_t *out);

/* z-depth (metres, +inf for a miss) of one pixel's ray; the scalar the counts
 * are floored from.  Used by the tests to tell quantisation ties from errors. */
double ora_render_pixel_depth(const ora_camera *cam, const float *triangles, int64_t n_tri,
                              const double cam_pos[3], const double att[4], const double mount[4], int px,
                              int py);

#ifdef __cplusplus
}
#endif
#endif
