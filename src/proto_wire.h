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
inline void put_varint(std::string &s, uint64_t v) {
  while (v >= 0x80) {
    s.push_back((char)((v & 0x7f) | 0x80));
    v >>= 7;
  }
  s.push_back((char)v);
}
inline void put_double(std::string &s, int field, double d) {
  uint64_t bits;
  std::memcpy(&bits, &d, 8);
  if (bits == 0) return;  // proto3: default values are not serialised (-0.0 has a non-zero pattern and is)
  put_varint(s, (uint64_t)(field << 3) | 1);
  s.append((const char *)&d, 8);
}
inline void put_msg(std::string &s, int field, const std::string &body) {
  put_varint(s, (uint64_t)(field << 3) | 2);
  put_varint(s, body.size());
  s.append(body);
}
inline std::string enc_vec(const double *v, int n) {
  std::string s;
  for (int i = 0; i < n; ++i) put_double(s, i + 1, v[i]);
  return s;
}
inline std::string enc_point(const double k[18]) {
  std::string so3, se3, st, pt;
  put_msg(so3, 1, enc_vec(k + 4, 4));
  put_msg(se3, 1, enc_vec(k + 1, 3));
  put_msg(se3, 2, so3);
  put_msg(st, 1, se3);
  put_msg(st, 2, enc_vec(k + 8, 6));
  put_double(pt, 1, k[0]);
  put_msg(pt, 2, st);
  put_msg(pt, 3, enc_vec(k + 14, 4));
  return pt;
}
inline std::string encode_trajectory(const double *traj, int n) {
  std::string s;
  for (int i = 0; i < n; ++i) put_msg(s, 1, enc_point(traj + (size_t)i * 18));
  return s;
}
// QuadrotorILQRDebug {repeated QuadrotorILQRIterDebug {trajectory = 1; cost = 2} iter_debugs = 1}
inline std::string encode_debug(const double *trajs, const double *costs, int n_iter, int n) {
  std::string s;
  for (int it = 0; it < n_iter; ++it) {
    std::string d;
    put_msg(d, 1, encode_trajectory(trajs + (size_t)it * n * 18, n));
    put_double(d, 2, costs[it]);
    put_msg(s, 1, d);
  }
  return s;
}

}  // namespace wire
