// pybind_module.cpp -- Python module `fcc_qp_solver`: the reference's Python surface
// (reference src/main.cpp:19-56: FCCQPDetails, FCCQPOptions, FCCQPSolution, FCCQP with the same
// attribute / method / keyword names) bound to the MI355X solver, plus FCCQPBatch.
//
// numpy in, numpy out.  Like the reference's Eigen::Ref casters, any array convertible to float64
// is accepted (C- or F-ordered, other dtypes are converted); friction_coeffs may be a list or an
// array.  Unlike the reference (whose size checks are asserts compiled out of its Release build),
// shape mismatches raise ValueError and too few friction coefficients raise IndexError.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "fcc_qp.hpp"
#include "fccqp.h"

namespace py = pybind11;

using fcc_qp::ConstMatrixView;
using fcc_qp::ConstVectorView;
using fcc_qp::FCCQP;
using fcc_qp::FCCQPBatch;
using fcc_qp::FCCQPBatchSolution;
using fcc_qp::FCCQPDetails;
using fcc_qp::FCCQPOptions;
using fcc_qp::FCCQPSolution;

namespace {

using darray = py::array_t<double, py::array::forcecast>;  // any layout; converts dtype
using carray = py::array_t<double, py::array::c_style | py::array::forcecast>;

ConstMatrixView mat_view(const darray &a, const char *name) {
  if (a.ndim() != 2) throw py::value_error(std::string(name) + " must be a 2-D array");
  return ConstMatrixView(a.data(), (long)a.shape(0), (long)a.shape(1),
                         (long)(a.strides(0) / (py::ssize_t)sizeof(double)),
                         (long)(a.strides(1) / (py::ssize_t)sizeof(double)));
}
ConstVectorView vec_view(const darray &a, const char *name) {
  if (a.ndim() == 2 && (a.shape(0) == 1 || a.shape(1) == 1)) {  // column / row vectors, as Eigen accepts
    const bool col = a.shape(1) == 1;
    return ConstVectorView(a.data(), (long)(col ? a.shape(0) : a.shape(1)),
                           (long)(a.strides(col ? 0 : 1) / (py::ssize_t)sizeof(double)));
  }
  if (a.ndim() != 1) throw py::value_error(std::string(name) + " must be a 1-D array");
  return ConstVectorView(a.data(), (long)a.shape(0), (long)(a.strides(0) / (py::ssize_t)sizeof(double)));
}

py::array_t<double> to_numpy(const std::vector<double> &v) {
  py::array_t<double> out((py::ssize_t)v.size());
  std::copy(v.begin(), v.end(), out.mutable_data());
  return out;
}

// a float64, C-contiguous array of exactly this shape, borrowed as it is (asynchronous entry points: no converted copy
// may be made, it would be gone before the device has read it)
const double *borrowed_f64(const py::array &a, std::initializer_list<py::ssize_t> shape, const char *name) {
  bool ok = a.dtype().is(py::dtype::of<double>()) && (a.flags() & py::array::c_style) && a.ndim() == (py::ssize_t)shape.size();
  if (ok) {
    int k = 0;
    for (auto d : shape) ok = ok && a.shape(k++) == d;
  }
  if (!ok) {
    std::string s = std::string(name) + " must be a C-contiguous float64 array of shape (";
    for (auto d : shape) s += std::to_string(d) + ",";
    throw py::value_error(s + ")");
  }
  return static_cast<const double *>(a.data());
}

template <class T>
void expect_shape_t(const py::array_t<T, py::array::c_style | py::array::forcecast> &a,
                    std::initializer_list<py::ssize_t> shape, const char *name) {
  bool ok = a.ndim() == (py::ssize_t)shape.size();
  if (ok) {
    int k = 0;
    for (auto d : shape) ok = ok && a.shape(k++) == d;
  }
  if (!ok) {
    std::string s = std::string(name) + " has the wrong shape; expected (";
    for (auto d : shape) s += std::to_string(d) + ",";
    s += ")";
    throw py::value_error(s);
  }
}

// FCCQPBatch.Solve for element type T (double, or float for the fp32-storage path)
template <class T>
void batch_solve(FCCQPBatch &self, const py::object &Qo, const py::object &bo, const py::object &Ao,
                 const py::object &beo, const py::object &fo, const py::object &lo, const py::object &uo) {
  using arr = py::array_t<T, py::array::c_style | py::array::forcecast>;
  const arr Q = arr::ensure(Qo), b = arr::ensure(bo), A_eq = arr::ensure(Ao), b_eq = arr::ensure(beo),
            friction_coeffs = arr::ensure(fo), lb = arr::ensure(lo), ub = arr::ensure(uo);
  if (!Q || !b || !A_eq || !b_eq || !friction_coeffs || !lb || !ub)
    throw py::type_error("FCCQPBatch.Solve: arguments must be convertible to numeric arrays");
  const py::ssize_t B = self.batch(), n = self.num_vars(), mm = self.num_equality_constraints();
  const py::ssize_t nf = self.nc() / 3;
  expect_shape_t<T>(Q, {B, n, n}, "Q");
  expect_shape_t<T>(b, {B, n}, "b");
  if (mm > 0) expect_shape_t<T>(A_eq, {B, mm, n}, "A_eq");
  if (mm > 0) expect_shape_t<T>(b_eq, {B, mm}, "b_eq");
  expect_shape_t<T>(lb, {B, n}, "lb");
  expect_shape_t<T>(ub, {B, n}, "ub");
  // friction_coeffs: [B, k] or a single [k] row shared by every instance, k >= nc/3
  std::vector<T> tiled;
  const T *fp = friction_coeffs.data();
  int k = 0;
  if (friction_coeffs.ndim() == 2 && friction_coeffs.shape(0) == B) {
    k = (int)friction_coeffs.shape(1);
  } else if (friction_coeffs.ndim() == 1) {
    k = (int)friction_coeffs.shape(0);
    tiled.resize((size_t)B * (size_t)k);
    for (py::ssize_t i = 0; i < B; ++i) std::copy(fp, fp + k, tiled.begin() + (size_t)i * (size_t)k);
    fp = tiled.data();
  } else {
    throw py::value_error("friction_coeffs must have shape (batch, k) or (k,)");
  }
  if (k < nf) throw py::index_error("friction_coeffs has fewer than nc/3 entries per instance");
  py::gil_scoped_release release;
  self.Solve(Q.data(), b.data(), A_eq.data(), b_eq.data(), fp, k, lb.data(), ub.data());
}

template <class T>
py::array_t<T> column(const FCCQPBatchSolution &s, T FCCQPDetails::*member) {
  py::array_t<T> out((py::ssize_t)s.batch);
  T *p = out.mutable_data();
  for (int i = 0; i < s.batch; ++i) p[i] = s.details[(size_t)i].*member;
  return out;
}

}  // namespace

PYBIND11_MODULE(fcc_qp_solver, m) {
  m.doc() = "Python bindings for the MI355X (HIP/gfx950) batched FCCQP solver for WBC";

  py::register_exception<fcc_qp::FccqpError>(m, "FCCQPError", PyExc_RuntimeError);

  // reference src/main.cpp:22-29 (no constructor; same attribute names)
  py::class_<FCCQPDetails>(m, "FCCQPDetails")
      .def_readwrite("n_iter", &FCCQPDetails::n_iter)
      .def_readwrite("eps_bounds", &FCCQPDetails::admm_residual_bounds)
      .def_readwrite("eps_friction_cone", &FCCQPDetails::admm_residual_friction_cone)
      .def_readwrite("bounds_viol", &FCCQPDetails::bounds_viol)
      .def_readwrite("friction_cone_viol", &FCCQPDetails::friction_cone_viol)
      .def_readwrite("solve_time", &FCCQPDetails::solve_time)
      .def_readwrite("factorization_time", &FCCQPDetails::factorization_time)
      // additions (C++-only in the reference: solve_status, src/fcc_qp.hpp:27)
      .def_property_readonly("solve_status", [](const FCCQPDetails &d) { return (int)d.solve_status; })
      .def_readonly("flags", &FCCQPDetails::flags);

  // reference src/main.cpp:31-36
  py::class_<FCCQPOptions>(m, "FCCQPOptions")
      .def(py::init<>())
      .def_readwrite("max_iter", &FCCQPOptions::max_iter)
      .def_readwrite("rho", &FCCQPOptions::rho)
      .def_readwrite("eps_fcone", &FCCQPOptions::eps_fcone)
      .def_readwrite("eps_bound", &FCCQPOptions::eps_bound);

  // reference src/main.cpp:38-40
  py::class_<FCCQPSolution>(m, "FCCQPSolution")
      .def_readwrite("details", &FCCQPSolution::details)
      .def_property(
          "z", [](const FCCQPSolution &s) { return to_numpy(s.z); },
          [](FCCQPSolution &s, const carray &v) { s.z.assign(v.data(), v.data() + v.size()); });

  // reference src/main.cpp:42-54
  py::class_<FCCQP>(m, "FCCQP")
      .def(py::init<int, int, int, int>(), py::arg("num_vars"), py::arg("num_equality_constraints"),
           py::arg("nc"), py::arg("lambda_c_start"))
      .def(py::init<int, int, int, int, int>(), py::arg("num_vars"),
           py::arg("num_equality_constraints"), py::arg("nc"), py::arg("lambda_c_start"),
           py::arg("device"))
      .def("set_rho", &FCCQP::set_rho)
      .def("set_max_iter", &FCCQP::set_max_iter)
      .def("set_warm_start", &FCCQP::set_warm_start)
      .def("set_options", &FCCQP::set_options)
      .def(
          "Solve",
          [](FCCQP &self, const darray &Q, const darray &b, const darray &A_eq, const darray &b_eq,
             const std::vector<double> &friction_coeffs, const darray &lb, const darray &ub) {
            ConstMatrixView vQ = mat_view(Q, "Q");
            ConstMatrixView vA = (A_eq.ndim() == 2) ? mat_view(A_eq, "A_eq")
                                                    : ConstMatrixView(A_eq.data(), 0, 0, 0, 0);
            if (A_eq.ndim() != 2 && A_eq.size() != 0) throw py::value_error("A_eq must be a 2-D array");
            self.Solve(vQ, vec_view(b, "b"), vA, vec_view(b_eq, "b_eq"), friction_coeffs,
                       vec_view(lb, "lb"), vec_view(ub, "ub"));
          },
          py::arg("Q"), py::arg("b"), py::arg("A_eq"), py::arg("b_eq"), py::arg("friction_coeffs"),
          py::arg("lb"), py::arg("ub"))
      .def("GetSolution", &FCCQP::GetSolution)
      .def("contact_vars_start", &FCCQP::contact_vars_start);

  // ---- batched API (new) ---------------------------------------------------------------
  py::class_<FCCQPBatchSolution>(m, "FCCQPBatchSolution")
      .def_property_readonly("z",
                             [](const FCCQPBatchSolution &s) {
                               py::array_t<double> out({(py::ssize_t)s.batch, (py::ssize_t)s.num_vars});
                               std::copy(s.z.begin(), s.z.end(), out.mutable_data());
                               return out;
                             })
      .def_property_readonly("n_iter", [](const FCCQPBatchSolution &s) { return column<int>(s, &FCCQPDetails::n_iter); })
      .def_property_readonly("eps_bounds", [](const FCCQPBatchSolution &s) { return column<double>(s, &FCCQPDetails::admm_residual_bounds); })
      .def_property_readonly("eps_friction_cone", [](const FCCQPBatchSolution &s) { return column<double>(s, &FCCQPDetails::admm_residual_friction_cone); })
      .def_property_readonly("bounds_viol", [](const FCCQPBatchSolution &s) { return column<double>(s, &FCCQPDetails::bounds_viol); })
      .def_property_readonly("friction_cone_viol", [](const FCCQPBatchSolution &s) { return column<double>(s, &FCCQPDetails::friction_cone_viol); })
      .def_property_readonly("solve_time", [](const FCCQPBatchSolution &s) { return column<double>(s, &FCCQPDetails::solve_time); })
      .def_property_readonly("factorization_time", [](const FCCQPBatchSolution &s) { return column<double>(s, &FCCQPDetails::factorization_time); })
      .def_property_readonly("flags", [](const FCCQPBatchSolution &s) { return column<int>(s, &FCCQPDetails::flags); })
      .def_property_readonly("solve_status",
                             [](const FCCQPBatchSolution &s) {
                               py::array_t<int> out((py::ssize_t)s.batch);
                               for (int i = 0; i < s.batch; ++i) out.mutable_data()[i] = (int)s.details[(size_t)i].solve_status;
                               return out;
                             })
      .def("__len__", [](const FCCQPBatchSolution &s) { return s.batch; })
      .def("__getitem__", [](const FCCQPBatchSolution &s, int i) {
        if (i < 0) i += s.batch;
        if (i < 0 || i >= s.batch) throw py::index_error();
        FCCQPSolution one;
        one.details = s.details[(size_t)i];
        one.z.assign(s.z.begin() + (size_t)i * s.num_vars, s.z.begin() + (size_t)(i + 1) * s.num_vars);
        return one;
      });

  py::class_<FCCQPBatch>(m, "FCCQPBatch")
      .def(py::init<int, int, int, int, int, int>(), py::arg("num_vars"),
           py::arg("num_equality_constraints"), py::arg("nc"), py::arg("lambda_c_start"),
           py::arg("batch"), py::arg("device") = 0)
      .def("set_rho", &FCCQPBatch::set_rho)
      .def("set_max_iter", &FCCQPBatch::set_max_iter)
      .def("set_warm_start", &FCCQPBatch::set_warm_start)
      .def("set_options", &FCCQPBatch::set_options)
      .def("set_kernel", &FCCQPBatch::set_kernel)
      .def("set_kernel_timing", &FCCQPBatch::set_kernel_timing, py::arg("enabled"))
      .def("kernel_in_use", &FCCQPBatch::kernel_in_use)
      .def("kernel_origin", [](const FCCQPBatch &self) {
        int o = 0;
        fccqp_get_kernel_origin(self.handle(), &o);
        return o;
      }, "0: the specialised kernel of this shape is compiled into the library (or there is none and none is needed), "
         "1: compiled at run time when the handle was created (include/fccqp.h, FCCQP_ORIGIN_*), 2: generic kernel only")
      .def_property_readonly("batch", &FCCQPBatch::batch)
      .def_property_readonly("num_vars", &FCCQPBatch::num_vars)
      .def_property_readonly("num_equality_constraints", &FCCQPBatch::num_equality_constraints)
      .def_property_readonly("nc", &FCCQPBatch::nc)
      .def_property_readonly("device", [](const FCCQPBatch &self) {
        int dev = 0;
        fccqp_get_shape(self.handle(), nullptr, nullptr, nullptr, nullptr, nullptr, &dev);
        return dev;
      })
      .def(
          "Solve",
          [](FCCQPBatch &self, const py::object &Q, const py::object &b, const py::object &A_eq,
             const py::object &b_eq, const py::object &friction_coeffs, const py::object &lb,
             const py::object &ub) {
            // a float32 Q selects the fp32-storage path (every argument is then taken as float32;
            // the kernels widen on load and compute in fp64); anything else is converted to float64
            const py::array qa = py::array::ensure(Q);
            if (qa && qa.dtype().is(py::dtype::of<float>()))
              batch_solve<float>(self, Q, b, A_eq, b_eq, friction_coeffs, lb, ub);
            else
              batch_solve<double>(self, Q, b, A_eq, b_eq, friction_coeffs, lb, ub);
          },
          py::arg("Q"), py::arg("b"), py::arg("A_eq"), py::arg("b_eq"), py::arg("friction_coeffs"),
          py::arg("lb"), py::arg("ub"))
      .def(
          "SolveDevice",
          [](FCCQPBatch &self, std::uintptr_t Q, std::uintptr_t b, std::uintptr_t A_eq, std::uintptr_t b_eq,
             std::uintptr_t friction_coeffs, std::uintptr_t lb, std::uintptr_t ub, std::uintptr_t stream,
             bool float32) {
            if (float32)
              self.SolveDevice(reinterpret_cast<const float *>(Q), reinterpret_cast<const float *>(b),
                               reinterpret_cast<const float *>(A_eq), reinterpret_cast<const float *>(b_eq),
                               reinterpret_cast<const float *>(friction_coeffs),
                               reinterpret_cast<const float *>(lb), reinterpret_cast<const float *>(ub),
                               reinterpret_cast<void *>(stream));
            else
              self.SolveDevice(reinterpret_cast<const double *>(Q), reinterpret_cast<const double *>(b),
                               reinterpret_cast<const double *>(A_eq), reinterpret_cast<const double *>(b_eq),
                               reinterpret_cast<const double *>(friction_coeffs),
                               reinterpret_cast<const double *>(lb), reinterpret_cast<const double *>(ub),
                               reinterpret_cast<void *>(stream));
          },
          py::arg("Q"), py::arg("b"), py::arg("A_eq"), py::arg("b_eq"), py::arg("friction_coeffs"),
          py::arg("lb"), py::arg("ub"), py::arg("stream") = 0, py::arg("float32") = false,
          "Asynchronous solve from device pointers (ints, e.g. torch.Tensor.data_ptr()) of "
          "instance-major float64 (or, with float32=True, float32) arrays resident on the solver's GPU.")
      .def("Synchronize", &FCCQPBatch::Synchronize, py::call_guard<py::gil_scoped_release>())
      .def("GetSolution", &FCCQPBatch::GetSolution)
      .def("LastKernelMilliseconds", &FCCQPBatch::LastKernelMilliseconds)
      .def("LastTiming",
           [](const FCCQPBatch &self) {
             const auto t = self.LastTiming();
             const char *by = t.second == FCCQP_TIMING_DEVICE_EVENTS ? "device_events"
                              : (t.second == FCCQP_TIMING_HOST_CLOCK ? "host_clock" : "none");
             return py::make_tuple(t.first, std::string(by));
           },
           "(milliseconds, how they were measured) of the last solve: 'device_events' (HIP events around the launches), "
           "'host_clock' (small synchronous host-pointer solves: launch to completion, results copied) or 'none' "
           "(kernel timing switched off)")
      .def("set_self_check", &FCCQPBatch::set_self_check, py::arg("enabled"),
           "re-check A_eq z = b_eq behind every solve (flag FLAG_EQ_RESIDUAL); default: on for num_vars + "
           "num_equality_constraints > 256")
      .def("set_dispatch_order_by_previous_iterations", &FCCQPBatch::set_dispatch_order_by_previous_iterations, py::arg("enabled"),
           "fccqp_set_dispatch_order: block -> instance order of the next launch of a chain, longest instance first by the "
           "previous solve's iteration counts (opt-in; results never depend on it)")
      .def("AlgorithmicBytes", &FCCQPBatch::AlgorithmicBytes)
      .def("device_pointers",
           [](FCCQPBatch &self) {
             const double *z = nullptr, *met = nullptr;
             const int *it = nullptr, *fl = nullptr;
             const long long *tk = nullptr;
             int rc = fccqp_get_solution_device(self.handle(), &z, &it, &fl, &met, &tk);
             if (rc != FCCQP_OK) throw fcc_qp::FccqpError(rc, fccqp_strerror(rc));
             py::dict d;
             d["z"] = (std::uintptr_t)z;
             d["n_iter"] = (std::uintptr_t)it;
             d["flags"] = (std::uintptr_t)fl;
             d["metrics"] = (std::uintptr_t)met;
             d["factor_ticks"] = (std::uintptr_t)tk;
             return d;
           },
           "Device addresses of the last solve's results (see include/fccqp.h).");

  // ---- one host batch over several GPUs (new) --------------------------------
  py::class_<fcc_qp::FCCQPSharded>(m, "FCCQPSharded")
      .def(py::init<int, int, int, int, int, const std::vector<int> &>(), py::arg("num_vars"),
           py::arg("num_equality_constraints"), py::arg("nc"), py::arg("lambda_c_start"),
           py::arg("batch"), py::arg("devices"))
      .def("set_options", &fcc_qp::FCCQPSharded::set_options)
      .def("set_warm_start", &fcc_qp::FCCQPSharded::set_warm_start)
      .def("set_kernel", &fcc_qp::FCCQPSharded::set_kernel)
      .def("set_kernel_timing", &fcc_qp::FCCQPSharded::set_kernel_timing, py::arg("enabled"))
      .def_property_readonly("batch", &fcc_qp::FCCQPSharded::batch)
      .def_property_readonly("shards", &fcc_qp::FCCQPSharded::shards)
      .def_property_readonly("num_vars", &fcc_qp::FCCQPSharded::num_vars)
      .def_property_readonly("num_equality_constraints", &fcc_qp::FCCQPSharded::num_equality_constraints)
      .def_property_readonly("nc", &fcc_qp::FCCQPSharded::nc)
      .def("shard_begin", &fcc_qp::FCCQPSharded::shard_begin)
      .def("device_of", &fcc_qp::FCCQPSharded::device_of)
      .def(
          "Solve",
          [](fcc_qp::FCCQPSharded &self, const py::object &Qo, const py::object &bo, const py::object &Ao,
             const py::object &beo, const py::object &fo, const py::object &lo, const py::object &uo) {
            using arr = py::array_t<double, py::array::c_style | py::array::forcecast>;
            const arr Q = arr::ensure(Qo), b = arr::ensure(bo), A_eq = arr::ensure(Ao), b_eq = arr::ensure(beo),
                      fr = arr::ensure(fo), lb = arr::ensure(lo), ub = arr::ensure(uo);
            if (!Q || !b || !A_eq || !b_eq || !fr || !lb || !ub)
              throw py::type_error("FCCQPSharded.Solve: arguments must be convertible to float64 arrays");
            const py::ssize_t B = self.batch(), n = self.num_vars(), mm = self.num_equality_constraints();
            expect_shape_t<double>(Q, {B, n, n}, "Q");
            expect_shape_t<double>(b, {B, n}, "b");
            if (mm > 0) expect_shape_t<double>(A_eq, {B, mm, n}, "A_eq");
            if (mm > 0) expect_shape_t<double>(b_eq, {B, mm}, "b_eq");
            expect_shape_t<double>(lb, {B, n}, "lb");
            expect_shape_t<double>(ub, {B, n}, "ub");
            if (!(fr.ndim() == 2 && fr.shape(0) == B)) throw py::value_error("friction_coeffs must have shape (batch, k)");
            const int k = (int)fr.shape(1);
            if (k < self.nc() / 3) throw py::index_error("friction_coeffs has fewer than nc/3 entries per instance");
            py::gil_scoped_release release;
            self.Solve(Q.data(), b.data(), A_eq.data(), b_eq.data(), fr.data(), k, lb.data(), ub.data());
          },
          py::arg("Q"), py::arg("b"), py::arg("A_eq"), py::arg("b_eq"), py::arg("friction_coeffs"),
          py::arg("lb"), py::arg("ub"),
          "Splits the instance-major host batch over the shards (one host thread per shard), solves, waits.")
      .def(
          "SolveDevice",
          [](fcc_qp::FCCQPSharded &self, const std::vector<std::vector<std::uintptr_t>> &shards, const std::vector<std::uintptr_t> &streams) {
            if ((int)shards.size() != self.shards()) throw py::value_error("FCCQPSharded.SolveDevice: one list of seven device pointers per shard");
            if (!streams.empty() && (int)streams.size() != self.shards()) throw py::value_error("FCCQPSharded.SolveDevice: one stream per shard, or none");
            std::vector<fcc_qp::FCCQPSharded::ShardDeviceInputs> in(shards.size());
            for (size_t g = 0; g < shards.size(); ++g) {
              if (shards[g].size() != 7) throw py::value_error("FCCQPSharded.SolveDevice: seven pointers per shard (Q, b, A_eq, b_eq, friction_coeffs, lb, ub)");
              const auto p = [&](int i) { return reinterpret_cast<const double *>(shards[g][(size_t)i]); };
              in[g].Q = p(0); in[g].b = p(1); in[g].A_eq = p(2); in[g].b_eq = p(3); in[g].friction_coeffs = p(4); in[g].lb = p(5); in[g].ub = p(6);
              if (!streams.empty()) in[g].stream = reinterpret_cast<void *>(streams[g]);
            }
            self.SolveDevice(in);
          },
          py::arg("shards"), py::arg("streams") = std::vector<std::uintptr_t>{},
          "Asynchronous solve from device-resident float64 arrays: shards[g] = the seven device pointers (ints) of shard g's "
          "instance-major slice, resident on device_of(g); streams[g] = the HIP stream to enqueue on (default: the shard's own).")
      .def(
          "SolveAsync",
          [](fcc_qp::FCCQPSharded &self, const py::array &Q, const py::array &b, const py::array &A_eq, const py::array &b_eq,
             const py::array &fr, const py::array &lb, const py::array &ub) {
            const py::ssize_t B = self.batch(), n = self.num_vars(), mm = self.num_equality_constraints(), nf = self.nc() / 3;
            const double *pQ = borrowed_f64(Q, {B, n, n}, "Q"), *pb = borrowed_f64(b, {B, n}, "b");
            const double *pA = mm > 0 ? borrowed_f64(A_eq, {B, mm, n}, "A_eq") : nullptr;
            const double *pbe = mm > 0 ? borrowed_f64(b_eq, {B, mm}, "b_eq") : nullptr;
            const double *pf = nf > 0 ? borrowed_f64(fr, {B, nf}, "friction_coeffs") : nullptr;
            const double *pl = borrowed_f64(lb, {B, n}, "lb"), *pu = borrowed_f64(ub, {B, n}, "ub");
            py::gil_scoped_release release;
            self.SolveAsync(pQ, pb, pA, pbe, pf, (int)nf, pl, pu);
          },
          py::arg("Q"), py::arg("b"), py::arg("A_eq"), py::arg("b_eq"), py::arg("friction_coeffs"), py::arg("lb"), py::arg("ub"),
          "Asynchronous solve from C-contiguous float64 host arrays (page-locked ones for copies that overlap): NO conversion "
          "is made (anything else is a ValueError), and the arrays must stay alive and unchanged until Synchronize().")
      .def(
          "GetZAsync",
          [](fcc_qp::FCCQPSharded &self, py::array &z) {
            if (!z.writeable()) throw py::value_error("z is read-only");
            self.GetZAsync(const_cast<double *>(borrowed_f64(z, {(py::ssize_t)self.batch(), (py::ssize_t)self.num_vars()}, "z")));
          },
          py::arg("z"), "Enqueues the copy of every shard's z into its rows of `z` ([batch, num_vars] float64, C-contiguous, kept alive until Synchronize()).")
      .def("Synchronize", &fcc_qp::FCCQPSharded::Synchronize, py::call_guard<py::gil_scoped_release>())
      .def("GetSolution", &fcc_qp::FCCQPSharded::GetSolution)
      .def("LastKernelMilliseconds", &fcc_qp::FCCQPSharded::LastKernelMilliseconds);

  m.def("device_count", &fccqp_device_count);
  m.def("kernel_for_shape",
        [](int num_vars, int num_equality_constraints, int nc, int lambda_c_start) {
          const int k = fccqp_kernel_for_shape(num_vars, num_equality_constraints, nc, lambda_c_start);
          if (k < 0) throw py::value_error(std::string("kernel_for_shape: ") + fccqp_strerror(-k));
          return k;
        },
        py::arg("num_vars"), py::arg("num_equality_constraints"), py::arg("nc"), py::arg("lambda_c_start"),
        "Which kernel KERNEL_AUTO takes for an FCCQP(num_vars, num_equality_constraints, nc, lambda_c_start) in this build: "
        "KERNEL_FUSED / KERNEL_COOP for a shape compiled in (fused_shapes.def / coop_shapes.def), KERNEL_GENERIC otherwise "
        "(several times slower; `python fcc_qp_amd/build.py --add-fused-shape n m nc s` adds it).  Needs no GPU.");
  m.attr("KERNEL_AUTO") = (int)FCCQP_KERNEL_AUTO;
  m.attr("KERNEL_GENERIC") = (int)FCCQP_KERNEL_GENERIC;
  m.attr("KERNEL_FUSED") = (int)FCCQP_KERNEL_FUSED;
  m.attr("KERNEL_FUSED_SPLIT") = (int)FCCQP_KERNEL_FUSED_SPLIT;
  m.attr("KERNEL_COOP") = (int)FCCQP_KERNEL_COOP;
  m.attr("KERNEL_COOP_SPLIT") = (int)FCCQP_KERNEL_COOP_SPLIT;
  m.attr("FLAG_PRESOLVE_RAN") = (int)FCCQP_FLAG_PRESOLVE_RAN;
  m.attr("FLAG_PRESOLVE_NUMERICAL") = (int)FCCQP_FLAG_PRESOLVE_NUMERICAL;
  m.attr("FLAG_ADMM_NUMERICAL") = (int)FCCQP_FLAG_ADMM_NUMERICAL;
  m.attr("FLAG_EQUALITY_ONLY") = (int)FCCQP_FLAG_EQUALITY_ONLY;
  m.attr("FLAG_PRESOLVE_COD") = (int)FCCQP_FLAG_PRESOLVE_COD;
  m.attr("FLAG_EQ_RESIDUAL") = (int)FCCQP_FLAG_EQ_RESIDUAL;
  m.attr("FLAG_PRESOLVE_CG") = (int)FCCQP_FLAG_PRESOLVE_CG;
}
