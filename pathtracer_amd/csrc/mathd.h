// FP64 vector/matrix helpers whose operation ORDER follows the reference's CGL
// library, because the order of roundings defines bit-exactness of the scene
// pipeline (CGL/include/CGL/vector3D.h:96-262, CGL/src/matrix3x3.cpp:110-114,
// CGL/src/matrix4x4.cpp:125-149, CGL/src/vector4D.cpp:10-17).
//
// Host code is compiled with -ffp-contract=off: the reference is built with
// -mavx2 and no -mfma, so none of its multiplies and adds are fused.
#pragma once
#include <cmath>

namespace ptamd {

constexpr double PI_D = 3.14159265358979323;  // CGL/include/CGL/misc.h:11
constexpr double EPS_D = 0.00000000001;        // CGL/include/CGL/misc.h:12
constexpr double EPS_F = 0.00001f;             // CGL/include/CGL/misc.h:13 (float literal widened)

struct V3 {
  double x, y, z;
  V3() : x(0), y(0), z(0) {}
  V3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
  double &operator[](int i) { return (&x)[i]; }
  const double &operator[](int i) const { return (&x)[i]; }
};

inline V3 operator+(const V3 &a, const V3 &b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(const V3 &a, const V3 &b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(const V3 &a) { return V3(-a.x, -a.y, -a.z); }
inline V3 operator*(const V3 &a, double c) { return V3(a.x * c, a.y * c, a.z * c); }
inline V3 operator*(double c, const V3 &a) { return V3(c * a.x, c * a.y, c * a.z); }
// vector / scalar multiplies by the reciprocal (vector3D.h:144-147)
inline V3 operator/(const V3 &a, double c) {
  const double rc = 1.0 / c;
  return V3(rc * a.x, rc * a.y, rc * a.z);
}
// (x*x' + y*y') + z*z' (vector3D.h:256-262; the SSE dppd path has the same order)
inline double dot(const V3 &u, const V3 &v) { return u.x * v.x + u.y * v.y + u.z * v.z; }
inline V3 cross(const V3 &u, const V3 &v) {
  return V3(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
inline double norm2(const V3 &a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
inline double norm(const V3 &a) { return std::sqrt(norm2(a)); }
// unit(): multiply by 1/norm (vector3D.h:215-218)
inline V3 unit(const V3 &a) {
  double r = 1. / norm(a);
  return a * r;
}
// normalize(): (*this) /= norm() -> *= (1./norm) (vector3D.h:159-167,223-225)
inline void normalize(V3 &a) {
  double r = 1. / norm(a);
  a.x *= r;
  a.y *= r;
  a.z *= r;
}

struct V4 {
  double x, y, z, w;
  V4() : x(0), y(0), z(0), w(0) {}
  V4(double x_, double y_, double z_, double w_) : x(x_), y(y_), z(z_), w(w_) {}
  V4(const V3 &v, double w_) : x(v.x), y(v.y), z(v.z), w(w_) {}
};
inline V4 operator*(double c, const V4 &v) { return V4(c * v.x, c * v.y, c * v.z, c * v.w); }
inline V4 operator+(const V4 &a, const V4 &b) { return V4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
inline double dot(const V4 &u, const V4 &v) { return u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w; }
inline V3 to3D(const V4 &v) { return V3(v.x, v.y, v.z); }
inline V3 projectTo3D(const V4 &v) {
  double invW = 1.0 / v.w;
  return V3(v.x * invW, v.y * invW, v.z * invW);
}

// Column-major 4x4; a default-constructed matrix is all zeros, as the
// reference's is (matrix4x4.h:28 leaves four zero-initialised Vector4D columns).
struct M4 {
  V4 col[4];
  double &operator()(int i, int j) { return (&col[j].x)[i]; }
  const double &operator()(int i, int j) const { return (&col[j].x)[i]; }
  static M4 identity() {
    M4 m;
    for (int i = 0; i < 4; i++) m(i, i) = 1.0;
    return m;
  }
};
inline V4 operator*(const M4 &m, const V4 &x) {
  return x.x * m.col[0] + x.y * m.col[1] + x.z * m.col[2] + x.w * m.col[3];
}
// The reference's AVX build computes C(i,j) = dot(column i of A, column j of B)
// (matrix4x4.cpp:131-132 indexes A[i], which is a column). This is A^T * B, not
// A * B; it is kept because it defines the node transforms the reference uses.
inline M4 mul_ref(const M4 &A, const M4 &B) {
  M4 C;
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) C(i, j) = dot(A.col[i], B.col[j]);
  return C;
}

// Column-major 3x3 (matrix3x3.cpp:110-114: M*v = v.x*col0 + v.y*col1 + v.z*col2).
struct M3 {
  V3 col[3];
};
inline V3 operator*(const M3 &m, const V3 &v) { return v.x * m.col[0] + v.y * m.col[1] + v.z * m.col[2]; }
inline M3 transpose(const M3 &a) {
  M3 b;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) b.col[j][i] = a.col[i][j];
  return b;
}

inline double radians(double deg) { return deg * (PI_D / 180); }  // misc.h:47-50
inline double degrees(double rad) { return rad * (180 / PI_D); }  // misc.h:55-58

}  // namespace ptamd
