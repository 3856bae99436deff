/*
 * maths.h — host-side vector types of the drop-in C++ API (vec3 / quat / mat3).
 *
 * Interface mirror of the reference's header of the same name (source/maths.h:23-104) so that
 * scene code written against the reference (`new Rigid(solver, {1,1,1}, 1.0f, 0.5f, {0,y,0}, quat())`,
 * `quat(axis, angle)`, `body->position.x`) compiles unchanged.  The arithmetic is delegated to
 * csrc/avbd_math.h — the same expression trees the device kernels use — so host-side set-up
 * values (masses, joint frames) are bit-identical to what the reference computes.
 */
#pragma once
#include "../csrc/avbd_math.h"

struct vec3 {
    float x, y, z;
    vec3() : x(0.0f), y(0.0f), z(0.0f) {}
    vec3(float s) : x(s), y(s), z(s) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    vec3(const avbd::f3& v) : x(v.x), y(v.y), z(v.z) {}
    operator avbd::f3() const { return avbd::mk3(x, y, z); }

    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }

    vec3 operator-() const { return vec3(-x, -y, -z); }
    vec3 operator+(const vec3& r) const { return vec3(x + r.x, y + r.y, z + r.z); }
    vec3 operator-(const vec3& r) const { return vec3(x - r.x, y - r.y, z - r.z); }
    vec3 operator*(float s) const { return vec3(x * s, y * s, z * s); }
    vec3 operator/(float s) const { return vec3(x / s, y / s, z / s); }
    vec3& operator+=(const vec3& r) { x += r.x; y += r.y; z += r.z; return *this; }
    vec3& operator-=(const vec3& r) { x -= r.x; y -= r.y; z -= r.z; return *this; }
    vec3& operator*=(float s) { x *= s; y *= s; z *= s; return *this; }
    vec3& operator/=(float s) { x /= s; y /= s; z /= s; return *this; }
};

inline vec3 operator*(float s, const vec3& v) { return v * s; }
inline float dot(const vec3& a, const vec3& b) { return avbd::dot3(a, b); }
inline float lengthSq(const vec3& v) { return avbd::lensq3(v); }
inline float length(const vec3& v) { return avbd::len3(v); }
inline vec3 cross(const vec3& a, const vec3& b) { return avbd::cross3(a, b); }
inline vec3 abs(const vec3& v) { return avbd::abs3(v); }
inline vec3 normalize(const vec3& v)
{
    float len = length(v);
    if (len < avbd::kVecEps) return vec3();
    return v / len;
}

struct quat {
    float x, y, z, w;
    quat() : x(0.0f), y(0.0f), z(0.0f), w(1.0f) {}
    quat(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
    quat(const avbd::q4& q) : x(q.x), y(q.y), z(q.z), w(q.w) {}
    /* axis-angle, maths.h:58 */
    quat(const vec3& axis, float angle)
    {
        float half = angle * 0.5f;
        float s = sinf(half);
        w = cosf(half);
        x = axis.x * s;
        y = axis.y * s;
        z = axis.z * s;
    }
    operator avbd::q4() const { return avbd::mkq(x, y, z, w); }
    quat operator+(const quat& r) const { return quat(x + r.x, y + r.y, z + r.z, w + r.w); }
    quat operator-(const quat& r) const { return quat(x - r.x, y - r.y, z - r.z, w - r.w); }
};

inline quat operator*(const quat& q, float s) { return quat(q.x * s, q.y * s, q.z * s, q.w * s); }
inline quat operator*(const quat& a, const quat& b) { return avbd::operator*(avbd::q4(a), avbd::q4(b)); }
inline quat normalize(const quat& q) { return avbd::qnormalize(q); }
inline quat conjugate(const quat& q) { return avbd::qconj(q); }
inline vec3 rotate(const quat& q, const vec3& v) { return avbd::qrotate(q, v); }

struct mat3 {
    vec3 cols[3];
    mat3() { cols[0] = vec3(1, 0, 0); cols[1] = vec3(0, 1, 0); cols[2] = vec3(0, 0, 1); }
    mat3(const vec3& c0, const vec3& c1, const vec3& c2) { cols[0] = c0; cols[1] = c1; cols[2] = c2; }
    mat3(const avbd::m33& m) { cols[0] = m.c0; cols[1] = m.c1; cols[2] = m.c2; }
    operator avbd::m33() const { return avbd::mkm(cols[0], cols[1], cols[2]); }
    static mat3 diagonal(float d) { return mat3(vec3(d, 0, 0), vec3(0, d, 0), vec3(0, 0, d)); }
    static mat3 diagonal(const vec3& v) { return mat3(vec3(v.x, 0, 0), vec3(0, v.y, 0), vec3(0, 0, v.z)); }
};

inline mat3 transpose(const mat3& m) { return avbd::mtranspose(m); }
inline vec3 operator*(const mat3& m, const vec3& v) { return avbd::operator*(avbd::m33(m), avbd::f3(v)); }
inline mat3 operator*(const mat3& a, const mat3& b) { return avbd::operator*(avbd::m33(a), avbd::m33(b)); }
inline mat3 operator+(const mat3& a, const mat3& b) { return avbd::operator+(avbd::m33(a), avbd::m33(b)); }
inline mat3 operator-(const mat3& a, const mat3& b) { return avbd::operator-(avbd::m33(a), avbd::m33(b)); }
inline mat3 operator*(const mat3& m, float s) { return avbd::operator*(avbd::m33(m), s); }
inline mat3 mat3_from_quat(const quat& q) { return avbd::quat_to_mat(q); }
inline vec3 solve(const mat3& A, const vec3& b) { return avbd::solve_ldlt3(A, b); }

inline float min(float a, float b) { return avbd::rmin(a, b); }
inline float max(float a, float b) { return avbd::rmax(a, b); }
inline float clamp(float x, float a, float b) { return avbd::rclamp(x, a, b); }
