// proto_wire.h -- minimal protobuf wire codec for the three message families of the reference's
// boundary (src/trajectory.proto, src/ilqr_options.proto, src/ilqr_debug.proto).  Replaces
// src/*_to_proto.{hh,cc} + libprotobuf + pybind11_protobuf, none of which exist in this image.
// Only wire types 0 (varint), 1 (fixed64) and 2 (length-delimited) occur in the schema; proto3
// omits zero-valued scalars, so absent fields decode to 0.  Unknown fields are skipped.
//
// A knot is 18 doubles in the order of include/quadrotor_ilqr.h:
//   [time_s, t(3), q(w,x,y,z), body_velocity(6), control(4)]
// The quaternion order on the wire is w,x,y,z (trajectory.proto:27-30, trajectory_to_proto.cc:67-83).
#pragma once
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace wire {

struct Reader {
  const uint8_t *p, *end;
  Reader(const uint8_t *b, size_t n) : p(b), end(b + n) {}
  bool done() const { return p >= end; }
  uint64_t varint() {
    uint64_t v = 0;
    for (int shift = 0; shift < 64; shift += 7) {
      if (p >= end) throw std::invalid_argument("protobuf: truncated varint");
      const uint8_t b = *p++;
      v |= (uint64_t)(b & 0x7f) << shift;
      if (!(b & 0x80)) return v;
    }
    throw std::invalid_argument("protobuf: varint too long");
  }
  double fixed64() {
    if (end - p < 8) throw std::invalid_argument("protobuf: truncated fixed64");
    double d;
    std::memcpy(&d, p, 8);  // little-endian host
    p += 8;
    return d;
  }
  Reader sub() {
    const uint64_t n = varint();
    if ((uint64_t)(end - p) < n) throw std::invalid_argument("protobuf: truncated message");
    Reader r(p, (size_t)n);
    p += n;
    return r;
  }
  void skip(int wt) {
    switch (wt) {
      case 0: varint(); break;
      case 1: if (end - p < 8) throw std::invalid_argument("protobuf: truncated"); p += 8; break;
      case 2: sub(); break;
      case 5: if (end - p < 4) throw std::invalid_argument("protobuf: truncated"); p += 4; break;
      default: throw std::invalid_argument("protobuf: unsupported wire type");
    }
  }
};

// VecN {c0 = 1 .. c(N-1) = N} doubles
inline void read_vec(Reader r, double *out, int n) {
  for (int i = 0; i < n; ++i) out[i] = 0.0;
  while (!r.done()) {
    const uint64_t tag = r.varint();
    const int f = (int)(tag >> 3), wt = (int)(tag & 7);
    if (wt == 1 && f >= 1 && f <= n) out[f - 1] = r.fixed64();
    else r.skip(wt);
  }
}
inline void read_so3(Reader r, double q[4]) {  // SO3 {Vec4 quaternion = 1}
  for (int i = 0; i < 4; ++i) q[i] = 0.0;
  while (!r.done()) {
    const uint64_t tag = r.varint();
    if ((tag >> 3) == 1 && (tag & 7) == 2) read_vec(r.sub(), q, 4);
    else r.skip((int)(tag & 7));
  }
}
inline void read_se3(Reader r, double pose[7]) {  // SE3 {Vec3 translation = 1; SO3 rotation = 2}
  for (int i = 0; i < 7; ++i) pose[i] = 0.0;
  while (!r.done()) {
    const uint64_t tag = r.varint();
    const int f = (int)(tag >> 3), wt = (int)(tag & 7);
    if (f == 1 && wt == 2) read_vec(r.sub(), pose, 3);
    else if (f == 2 && wt == 2) read_so3(r.sub(), pose + 3);
    else r.skip(wt);
  }
}
inline void read_state(Reader r, double x[13]) {  // QuadrotorState {SE3 = 1; Vec6 body_velocity = 2}
  for (int i = 0; i < 13; ++i) x[i] = 0.0;
  while (!r.done()) {
    const uint64_t tag = r.varint();
    const int f = (int)(tag >> 3), wt = (int)(tag & 7);
    if (f == 1 && wt == 2) read_se3(r.sub(), x);
    else if (f == 2 && wt == 2) read_vec(r.sub(), x + 7, 6);
    else r.skip(wt);
  }
}
inline void read_point(Reader r, double k[18]) {  // {time_s = 1; state = 2; control = 3}
  for (int i = 0; i < 18; ++i) k[i] = 0.0;
  while (!r.done()) {
    const uint64_t tag = r.varint();
    const int f = (int)(tag >> 3), wt = (int)(tag & 7);
    if (f == 1 && wt == 1) k[0] = r.fixed64();
    else if (f == 2 && wt == 2) read_state(r.sub(), k + 1);
    else if (f == 3 && wt == 2) read_vec(r.sub(), k + 14, 4);
    else r.skip(wt);
  }
}
// QuadrotorTrajectory {repeated QuadrotorTrajectoryPoint points = 1} -> n x 18
inline std::vector<double> decode_trajectory(const std::string &bytes) {
  Reader r((const uint8_t *)bytes.data(), bytes.size());
  std::vector<double> out;
  while (!r.done()) {
    const uint64_t tag = r.varint();
    if ((tag >> 3) == 1 && (tag & 7) == 2) {
      out.resize(out.size() + 18);
      read_point(r.sub(), out.data() + out.size() - 18);
    } else {
      r.skip((int)(tag & 7));
    }
  }
  return out;
}

struct Options {
  double step_update = 0, desired_reduction_frac = 0;
  int32_t ls_max_iters = 0;
  double rtol = 0, atol = 0, max_iters = 0;
  bool populate_debug = false;
};
// ILQROptions {LineSearchParams = 1; ConvergenceCriteria = 2; bool populate_debug = 3}
inline Options decode_options(const std::string &bytes) {
  Options o;
  Reader r((const uint8_t *)bytes.data(), bytes.size());
  while (!r.done()) {
    const uint64_t tag = r.varint();
    const int f = (int)(tag >> 3), wt = (int)(tag & 7);
    if (f == 1 && wt == 2) {
      Reader s = r.sub();
      while (!s.done()) {
        const uint64_t t2 = s.varint();
        const int f2 = (int)(t2 >> 3), w2 = (int)(t2 & 7);
        if (f2 == 1 && w2 == 1) o.step_update = s.fixed64();
        else if (f2 == 2 && w2 == 1) o.desired_reduction_frac = s.fixed64();
        else if (f2 == 3 && w2 == 0) o.ls_max_iters = (int32_t)s.varint();
        else s.skip(w2);
      }
    } else if (f == 2 && wt == 2) {
      Reader s = r.sub();
      while (!s.done()) {
        const uint64_t t2 = s.varint();
        const int f2 = (int)(t2 >> 3), w2 = (int)(t2 & 7);
        if (f2 == 1 && w2 == 1) o.rtol = s.fixed64();
        else if (f2 == 2 && w2 == 1) o.atol = s.fixed64();
        else if (f2 == 3 && w2 == 1) o.max_iters = s.fixed64();  // a double in the schema
        else s.skip(w2);
      }
    } else if (f == 3 && wt == 0) {
      o.populate_debug = r.varint() != 0;
    } else {
      r.skip(wt);
    }
  }
  return o;
}

// ------------------------------------------------------------------ encoding
// Two passes over the values -- sizes, then bytes into ONE buffer of the exact size (round 5: the first version built every nested
// message in a std::string of its own and copied it into its parent, ~ 10 allocations and 6 copies per knot; ILQRDebug of
// 100 iterations x 100 knots took 8.6 ms to encode, more than half of what populate_debug cost a single solve through the binding).
// proto3: zero-valued doubles are not serialised (-0.0 has a non-zero bit pattern and is); a sub-message is always written, empty or not
// (the reference's converters set every one).  Field numbers are below 16: one-byte tags.
inline bool nonzero(double d) {
  uint64_t bits;
  std::memcpy(&bits, &d, 8);
  return bits != 0;
}
inline size_t varint_len(uint64_t v) {
  size_t n = 1;
  while (v >= 0x80) { v >>= 7; ++n; }
  return n;
}
inline uint8_t *write_varint(uint8_t *p, uint64_t v) {
  while (v >= 0x80) {
    *p++ = (uint8_t)((v & 0x7f) | 0x80);
    v >>= 7;
  }
  *p++ = (uint8_t)v;
  return p;
}
inline uint8_t *write_double(uint8_t *p, int field, double d) {
  if (!nonzero(d)) return p;
  *p++ = (uint8_t)((field << 3) | 1);
  std::memcpy(p, &d, 8);
  return p + 8;
}
inline uint8_t *write_header(uint8_t *p, int field, size_t len) {  // tag of a length-delimited field and its length
  *p++ = (uint8_t)((field << 3) | 2);
  return write_varint(p, len);
}
inline size_t vec_size(const double *v, int n) {
  size_t s = 0;
  for (int i = 0; i < n; ++i) s += nonzero(v[i]) ? 9 : 0;
  return s;
}
inline uint8_t *write_vec(uint8_t *p, const double *v, int n) {
  for (int i = 0; i < n; ++i) p = write_double(p, i + 1, v[i]);
  return p;
}
inline size_t msg_size(size_t body) { return 1 + varint_len(body) + body; }  // a length-delimited field holding `body` bytes
struct PointSizes {
  size_t q, so3, t, se3, vel, state, u, point;
};
inline PointSizes point_sizes(const double k[18]) {
  PointSizes z;
  z.q = vec_size(k + 4, 4);
  z.so3 = msg_size(z.q);                       // SO3 {Vec4 quaternion = 1}
  z.t = vec_size(k + 1, 3);
  z.se3 = msg_size(z.t) + msg_size(z.so3);     // SE3 {translation = 1; rotation = 2}
  z.vel = vec_size(k + 8, 6);
  z.state = msg_size(z.se3) + msg_size(z.vel);  // QuadrotorState {inertial_from_body = 1; body_velocity = 2}
  z.u = vec_size(k + 14, 4);
  z.point = (nonzero(k[0]) ? 9 : 0) + msg_size(z.state) + msg_size(z.u);  // {time_s = 1; state = 2; control = 3}
  return z;
}
inline uint8_t *write_point(uint8_t *p, const double k[18], const PointSizes &z) {
  p = write_double(p, 1, k[0]);
  p = write_header(p, 2, z.state);
  p = write_header(p, 1, z.se3);
  p = write_header(p, 1, z.t);
  p = write_vec(p, k + 1, 3);
  p = write_header(p, 2, z.so3);
  p = write_header(p, 1, z.q);
  p = write_vec(p, k + 4, 4);
  p = write_header(p, 2, z.vel);
  p = write_vec(p, k + 8, 6);
  p = write_header(p, 3, z.u);
  return write_vec(p, k + 14, 4);
}
inline size_t trajectory_size(const double *traj, int n) {
  size_t s = 0;
  for (int i = 0; i < n; ++i) s += msg_size(point_sizes(traj + (size_t)i * 18).point);
  return s;
}
inline uint8_t *write_trajectory(uint8_t *p, const double *traj, int n) {
  for (int i = 0; i < n; ++i) {
    const double *k = traj + (size_t)i * 18;
    const PointSizes z = point_sizes(k);
    p = write_header(p, 1, z.point);
    p = write_point(p, k, z);
  }
  return p;
}
// QuadrotorTrajectory {repeated QuadrotorTrajectoryPoint points = 1}
inline std::string encode_trajectory(const double *traj, int n) {
  std::string s(trajectory_size(traj, n), '\0');
  uint8_t *end = write_trajectory((uint8_t *)&s[0], traj, n);
  if ((size_t)(end - (uint8_t *)&s[0]) != s.size()) throw std::logic_error("protobuf: size pass and write pass disagree");
  return s;
}
// QuadrotorILQRDebug {repeated QuadrotorILQRIterDebug {trajectory = 1; cost = 2} iter_debugs = 1}
inline std::string encode_debug(const double *trajs, const double *costs, int n_iter, int n) {
  std::vector<size_t> tsz((size_t)(n_iter > 0 ? n_iter : 0));
  size_t total = 0;
  for (int it = 0; it < n_iter; ++it) {
    tsz[it] = trajectory_size(trajs + (size_t)it * n * 18, n);
    total += msg_size(msg_size(tsz[it]) + (nonzero(costs[it]) ? 9 : 0));
  }
  std::string s(total, '\0');
  uint8_t *p = (uint8_t *)&s[0];
  for (int it = 0; it < n_iter; ++it) {
    p = write_header(p, 1, msg_size(tsz[it]) + (nonzero(costs[it]) ? 9 : 0));
    p = write_header(p, 1, tsz[it]);
    p = write_trajectory(p, trajs + (size_t)it * n * 18, n);
    p = write_double(p, 2, costs[it]);
  }
  if ((size_t)(p - (uint8_t *)&s[0]) != s.size()) throw std::logic_error("protobuf: size pass and write pass disagree");
  return s;
}

}  // namespace wire
