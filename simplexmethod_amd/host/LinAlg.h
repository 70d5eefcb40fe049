// LinAlg.h — the dense fp64 containers the problem classes exchange.
//
// The reference passes Eigen::MatrixXd / VectorXd / VectorXi (column-major, Eigen's
// default) through its class API (/root/reference/src/ProblemTypes/IProblem.h:3-15).
// Eigen is not vendored here (the reference fetches it at configure time,
// CMakeLists.txt:12-17), so when <Eigen/Dense> is on the include path these names ARE
// Eigen's types and the classes below interoperate with reference code unchanged;
// otherwise a minimal column-major implementation with the same member spelling is used.
// No arithmetic worth a GPU happens in these containers — the hot path hands their
// .data() pointers to the C ABI (include/simplexmethod_amd.h).
#pragma once

#if __has_include(<Eigen/Dense>) && !defined(LPLA_FORCE_BUILTIN)
#include <Eigen/Dense>
namespace lpla {
using MatrixXd = Eigen::MatrixXd;
using VectorXd = Eigen::VectorXd;
using VectorXi = Eigen::VectorXi;
constexpr bool kUsingEigen = true;
}  // namespace lpla
#else
#include <cstddef>
#include <initializer_list>
#include <stdexcept>
#include <vector>

namespace lpla {
constexpr bool kUsingEigen = false;

template <typename T>
class Vector {
public:
    Vector() = default;
    explicit Vector(long n) : v_((size_t)n, T(0)) {}
    Vector(std::initializer_list<T> il) : v_(il) {}
    static Vector Zero(long n) { return Vector(n); }
    long size() const { return (long)v_.size(); }
    void resize(long n) { v_.assign((size_t)n, T(0)); }
    T& operator()(long i) { return v_[(size_t)i]; }
    const T& operator()(long i) const { return v_[(size_t)i]; }
    T& operator[](long i) { return v_[(size_t)i]; }
    const T& operator[](long i) const { return v_[(size_t)i]; }
    T* data() { return v_.data(); }
    const T* data() const { return v_.data(); }
    Vector head(long n) const {
        Vector r(n);
        for (long i = 0; i < n; ++i) r[i] = v_[(size_t)i];
        return r;
    }
    T dot(const Vector& o) const {  // Eigen's c.dot(x), used by Evaluate (Canonical.cpp:86)
        if (o.size() != size()) throw std::invalid_argument("dot: size mismatch");
        T z = T(0);
        for (long i = 0; i < size(); ++i) z += v_[(size_t)i] * o[i];
        return z;
    }
    bool operator==(const Vector& o) const { return v_ == o.v_; }

private:
    std::vector<T> v_;
};

class MatrixXd {
public:
    MatrixXd() = default;
    MatrixXd(long r, long c) : r_(r), c_(c), v_((size_t)(r * c), 0.0) {}
    static MatrixXd Zero(long r, long c) { return MatrixXd(r, c); }
    static MatrixXd Identity(long r, long c) {
        MatrixXd m(r, c);
        for (long i = 0; i < (r < c ? r : c); ++i) m(i, i) = 1.0;
        return m;
    }
    // row-major initializer, mirroring Eigen's comma initializer used by the reference tests
    static MatrixXd FromRows(long r, long c, std::initializer_list<double> il) {
        if ((long)il.size() != r * c) throw std::invalid_argument("FromRows: wrong element count");
        MatrixXd m(r, c);
        long k = 0;
        for (double x : il) {
            m(k / c, k % c) = x;
            ++k;
        }
        return m;
    }
    long rows() const { return r_; }
    long cols() const { return c_; }
    void resize(long r, long c) {
        r_ = r;
        c_ = c;
        v_.assign((size_t)(r * c), 0.0);
    }
    double& operator()(long i, long j) { return v_[(size_t)(j * r_ + i)]; }   // column-major
    const double& operator()(long i, long j) const { return v_[(size_t)(j * r_ + i)]; }
    double* data() { return v_.data(); }
    const double* data() const { return v_.data(); }
    MatrixXd transpose() const {
        MatrixXd t(c_, r_);
        for (long j = 0; j < c_; ++j)
            for (long i = 0; i < r_; ++i) t(j, i) = (*this)(i, j);
        return t;
    }
    bool operator==(const MatrixXd& o) const { return r_ == o.r_ && c_ == o.c_ && v_ == o.v_; }

private:
    long r_ = 0, c_ = 0;
    std::vector<double> v_;
};

using VectorXd = Vector<double>;
using VectorXi = Vector<int>;
}  // namespace lpla
#endif
