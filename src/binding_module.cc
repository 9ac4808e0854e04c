// binding_module.cc -- Python module `src.quadrotor_ilqr_binding`: the reference's pybind11
// surface (src/quadrotor_ilqr_binding.cc:20-50) re-exposed over the C ABI of the HIP solver
// (include/quadrotor_ilqr.h).
//
//   QuadrotorILQR(mass_kg, inertia(3,3), arm_length_m, torque_to_thrust_ratio_m, g_mpss,
//                 Q(12,12), R(4,4), desired_traj: QuadrotorTrajectory, dt_s, options: ILQROptions)
//       positional only, no defaults                                  (binding.cc:20-32, 47-48)
//   .solve(initial_traj: QuadrotorTrajectory) -> (QuadrotorTrajectory, QuadrotorILQRDebug)
//                                                                     (binding.cc:34-41, 49)
// Messages cross as serialised bytes (what pybind11_protobuf does when the Python protobuf runtime
// is not the C++ one) through src/proto_wire.h.  Errors map as the reference's exceptions do under
// pybind11: runtime_error -> RuntimeError (inertia, line-search exhaustion with the same text),
// out_of_range -> IndexError, invalid_argument -> ValueError, wrong array shape -> TypeError.
#include <cmath>
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>

#include <memory>
#include <string>
#include <vector>

#include "../include/quadrotor_ilqr.h"
#include "proto_wire.h"

namespace py = pybind11;

namespace {

std::string to_bytes(const py::object &msg, const char *expected_type) {
  if (!py::hasattr(msg, "SerializeToString") || !py::hasattr(msg, "DESCRIPTOR"))
    throw py::type_error(std::string("expected a src.proto.") + expected_type + " message");
  const std::string full = py::str(msg.attr("DESCRIPTOR").attr("full_name"));
  if (full != std::string("src.proto.") + expected_type)
    throw py::type_error("expected src.proto." + std::string(expected_type) + ", got " + full);
  return py::bytes(msg.attr("SerializeToString")());
}

py::object from_bytes(const char *module, const char *cls, const std::string &bytes) {
  py::object klass = py::module_::import(module).attr(cls);
  return klass.attr("FromString")(py::bytes(bytes));
}

std::vector<double> matrix(const py::object &o, int rows, int cols, const char *name) {
  auto a = py::array_t<double, py::array::c_style | py::array::forcecast>::ensure(o);
  if (!a || a.ndim() != 2 || a.shape(0) != rows || a.shape(1) != cols)
    throw py::type_error(std::string(name) + " must be a float64 array of shape (" + std::to_string(rows) + ", " +
                         std::to_string(cols) + ")");
  return std::vector<double>(a.data(), a.data() + rows * cols);
}

[[noreturn]] void raise(int rc) {
  const std::string msg = qilqr_last_error();
  switch (rc) {
    case QILQR_ERR_BAD_INERTIA: throw std::runtime_error("Inertia matrix is not positive definite!");
    case QILQR_ERR_LENGTH_MISMATCH: throw py::index_error(msg);
    case QILQR_ERR_BAD_QUATERNION: throw py::value_error(msg);
    case QILQR_ERR_INVALID_ARG: throw py::value_error(msg);
    case QILQR_ERR_LINE_SEARCH: throw std::runtime_error(msg);
    default: throw std::runtime_error("quadrotor_ilqr: " + msg);
  }
}

struct Solver {
  qilqr_solver *h = nullptr;
  wire::Options opt;
  ~Solver() { qilqr_destroy(h); }
};

std::unique_ptr<Solver> init(double mass_kg, const py::object &inertia, double arm_length_m,
                             double torque_to_thrust_ratio_m, double g_mpss, const py::object &Q,
                             const py::object &R, const py::object &desired_traj, double dt_s,
                             const py::object &options) {
  const std::vector<double> I = matrix(inertia, 3, 3, "inertia");
  const std::vector<double> q = matrix(Q, 12, 12, "Q");
  const std::vector<double> r = matrix(R, 4, 4, "R");
  const std::vector<double> desired = wire::decode_trajectory(to_bytes(desired_traj, "QuadrotorTrajectory"));
  auto s = std::make_unique<Solver>();
  s->opt = wire::decode_options(to_bytes(options, "ILQROptions"));
  qilqr_model m;
  m.mass_kg = mass_kg;
  for (int i = 0; i < 9; ++i) m.inertia[i] = I[i];
  m.arm_length_m = arm_length_m;
  m.torque_to_thrust_ratio_m = torque_to_thrust_ratio_m;
  m.g_mpss = g_mpss;
  qilqr_options o;
  o.step_update = s->opt.step_update;
  o.desired_reduction_frac = s->opt.desired_reduction_frac;
  o.ls_max_iters = s->opt.ls_max_iters;
  o.rtol = s->opt.rtol;
  o.atol = s->opt.atol;
  o.max_iters = s->opt.max_iters;
  o.populate_debug = s->opt.populate_debug ? 1 : 0;
  const int rc = qilqr_create(&m, q.data(), r.data(), desired.data(), (int32_t)(desired.size() / 18), dt_s, &o,
                              nullptr, &s->h);
  if (rc) raise(rc);
  return s;
}

py::tuple solve(Solver &self, const py::object &initial_traj) {
  const std::vector<double> init = wire::decode_trajectory(to_bytes(initial_traj, "QuadrotorTrajectory"));
  const int n = (int)(init.size() / 18);
  if (n == 0) throw py::value_error("empty initial trajectory");  // UB in the reference (ilqr.hh:156)
  std::vector<double> out(init.size());
  int cap = 0;
  if (self.opt.populate_debug && self.opt.max_iters > 0) cap = (int)std::min(std::ceil(self.opt.max_iters), 1e6);  // i < max_iters, a double (ilqr.hh:58)
  std::vector<double> dcost((size_t)std::max(cap, 1)), dtraj((size_t)std::max(cap, 1) * init.size());
  double cost = 0;
  int32_t status = 0, iters = 0, n_debug = 0;
  int rc;
  {
    py::gil_scoped_release nogil;  // the handle is re-entrant; the reference holds the GIL
    rc = qilqr_solve(self.h, init.data(), n, out.data(), &cost, &status, &iters, dcost.data(), dtraj.data(), cap,
                     &n_debug);
  }
  if (rc) raise(rc);
  py::object traj = from_bytes("src.trajectory_pb2", "QuadrotorTrajectory", wire::encode_trajectory(out.data(), n));
  py::object debug = from_bytes("src.ilqr_debug_pb2", "QuadrotorILQRDebug",
                                wire::encode_debug(dtraj.data(), dcost.data(), n_debug, n));
  return py::make_tuple(traj, debug);
}

// test hooks: the codec against python-protobuf (trajectory_to_proto_test.cc:13-38,
// ilqr_options_to_proto_test.cc:7-18, ilqr_debug_to_proto_test.cc:30-41)
py::array_t<double> decode_trajectory_msg(const py::object &msg) {
  const std::vector<double> t = wire::decode_trajectory(to_bytes(msg, "QuadrotorTrajectory"));
  py::array_t<double> a({(py::ssize_t)(t.size() / 18), (py::ssize_t)18});
  std::memcpy(a.mutable_data(), t.data(), t.size() * sizeof(double));
  return a;
}
py::object encode_trajectory_arr(const py::object &arr) {
  auto a = py::array_t<double, py::array::c_style | py::array::forcecast>::ensure(arr);
  if (!a || a.ndim() != 2 || a.shape(1) != 18) throw py::type_error("expected an (n, 18) float64 array");
  return from_bytes("src.trajectory_pb2", "QuadrotorTrajectory", wire::encode_trajectory(a.data(), (int)a.shape(0)));
}
py::tuple decode_options_msg(const py::object &msg) {
  const wire::Options o = wire::decode_options(to_bytes(msg, "ILQROptions"));
  return py::make_tuple(o.step_update, o.desired_reduction_frac, o.ls_max_iters, o.rtol, o.atol, o.max_iters,
                        o.populate_debug);
}
py::object encode_debug_arr(const py::object &trajs, const py::object &costs) {
  auto t = py::array_t<double, py::array::c_style | py::array::forcecast>::ensure(trajs);
  auto c = py::array_t<double, py::array::c_style | py::array::forcecast>::ensure(costs);
  if (!t || !c || t.ndim() != 3 || t.shape(2) != 18 || c.ndim() != 1 || c.shape(0) != t.shape(0))
    throw py::type_error("expected (k, n, 18) trajectories and (k,) costs");
  return from_bytes("src.ilqr_debug_pb2", "QuadrotorILQRDebug",
                    wire::encode_debug(t.data(), c.data(), (int)t.shape(0), (int)t.shape(1)));
}

}  // namespace

PYBIND11_MODULE(quadrotor_ilqr_binding, m) {
  m.doc() = "MI355X-native drop-in for the reference's quadrotor_ilqr_binding (HIP solver behind a C ABI)";
  py::class_<Solver>(m, "QuadrotorILQR")
      .def(py::init(&init))  // positional only, like the reference: no argument names, no defaults
      .def("solve", &solve);
  m.def("_decode_trajectory", &decode_trajectory_msg);
  m.def("_encode_trajectory", &encode_trajectory_arr);
  m.def("_decode_options", &decode_options_msg);
  m.def("_encode_debug", &encode_debug_arr);
}
