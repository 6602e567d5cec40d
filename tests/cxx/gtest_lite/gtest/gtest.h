// gtest_lite — a single-header stand-in for the parts of GoogleTest the reference's unit tests use, so that those tests (test/utils,
// test/misc/ibm, test/misc/lanczos, test/BDHI/FCM, test/BDHI/PSE: written against <gtest/gtest.h> + <gmock/gmock.h>, which the reference
// fetches from the network at configure time, test/CMakeLists.txt:10-16) compile UNCHANGED against include/uammd and run on the GPU box.
// Own code (no GoogleTest source): TEST() / TEST_F() registration (fixtures derive from ::testing::Test: SetUp, TearDown), EXPECT_* / ASSERT_* with streamed messages, ASSERT_THAT / EXPECT_THAT with the
// DoubleNear matcher, gtest-style progress lines, --gtest_filter / --gtest_list_tests, exit status 1 when any test failed.  The header
// defines main() (the tests link gtest_main in the reference) unless GTEST_LITE_NO_MAIN is defined.
#ifndef UAMMD_TESTS_GTEST_LITE_H
#define UAMMD_TESTS_GTEST_LITE_H
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <exception>
#include <functional>
#include <iomanip>
#include <iostream>
#include <sstream>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

namespace testing {

// ---- streamed message: `ASSERT_EQ(a, b) << "context " << i;` -------------------------------------------------------------------------
class Message {
  std::ostringstream os;
public:
  Message() { os << std::setprecision(17); }
  Message(const Message &o) { os << o.os.str(); }
  template <class T> Message &operator<<(const T &v) { os << v; return *this; }
  Message &operator<<(std::ostream &(*manip)(std::ostream &)) { os << manip; return *this; }
  std::string str() const { return os.str(); }
};

namespace internal {
struct TestInfo { std::string suite, name; void (*body)(); };
inline std::vector<TestInfo> &registry() { static std::vector<TestInfo> r; return r; }
struct Registrar { Registrar(const char *s, const char *n, void (*b)()) { registry().push_back({s, n, b}); } };
struct State { int failuresInCurrentTest = 0; };
inline State &state() { static State s; return s; }

// prints a value when it can be streamed, its size otherwise
template <class T, class = void> struct Printer {
  static void print(std::ostream &o, const T &) { o << "<" << sizeof(T) << "-byte object>"; }
};
template <class T> struct Printer<T, decltype(void(std::declval<std::ostream &>() << std::declval<const T &>()))> {
  static void print(std::ostream &o, const T &v) { o << v; }
};
template <class T> std::string show(const T &v) { std::ostringstream o; o << std::setprecision(17); Printer<T>::print(o, v); return o.str(); }
inline std::string show(bool v) { return v ? "true" : "false"; }
inline std::string show(std::nullptr_t) { return "nullptr"; }

// `AssertHelper(...) = Message() << ...` : the message is complete when the assignment runs; the assignment reports the failure
class AssertHelper {
  const char *file; int line; std::string summary;
public:
  AssertHelper(const char *f, int l, std::string s) : file(f), line(l), summary(std::move(s)) {}
  void operator=(const Message &m) const {
    ++state().failuresInCurrentTest;
    std::cout << file << ":" << line << ": Failure\n" << summary;
    const std::string extra = m.str();
    if (!extra.empty()) std::cout << "\n" << extra;
    std::cout << std::endl;
  }
};
}  // namespace internal

class AssertionResult {
  bool ok;
  std::string msg;
public:
  AssertionResult(bool ok_, std::string m = std::string()) : ok(ok_), msg(std::move(m)) {}
  explicit operator bool() const { return ok; }
  const std::string &message() const { return msg; }
};
inline AssertionResult AssertionSuccess() { return AssertionResult(true); }
inline AssertionResult AssertionFailure() { return AssertionResult(false); }

namespace internal {
#define GTL_COMPARE_(Name, op)                                                                                          \
  template <class A, class B> AssertionResult Cmp##Name(const char *ea, const char *eb, const A &a, const B &b) {         \
    if (a op b) return AssertionSuccess();                                                                              \
    return AssertionResult(false, std::string("Expected: (") + ea + ") " #op " (" + eb + "), actual: " + show(a) + " vs " + show(b)); \
  }
#if defined(__clang__)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wsign-compare"
#elif defined(__GNUC__)
#pragma GCC diagnostic push
#pragma GCC diagnostic ignored "-Wsign-compare"
#endif
GTL_COMPARE_(EQ, ==)
GTL_COMPARE_(NE, !=)
GTL_COMPARE_(LT, <)
GTL_COMPARE_(LE, <=)
GTL_COMPARE_(GT, >)
GTL_COMPARE_(GE, >=)
#if defined(__clang__)
#pragma clang diagnostic pop
#elif defined(__GNUC__)
#pragma GCC diagnostic pop
#endif
#undef GTL_COMPARE_
inline AssertionResult CmpBool(const char *e, bool value, bool expected) {
  if (value == expected) return AssertionSuccess();
  return AssertionResult(false, std::string("Value of: ") + e + "\n  Actual: " + show(value) + "\nExpected: " + show(expected));
}
inline AssertionResult CmpNear(const char *ea, const char *eb, const char *et, double a, double b, double tol) {
  const double diff = std::fabs(a - b);
  if (diff <= tol) return AssertionSuccess();   // (a NaN on either side fails, as in GoogleTest)
  return AssertionResult(false, std::string("The difference between ") + ea + " and " + eb + " is " + show(diff) + ", which exceeds " + et +
                                    ", where\n" + ea + " evaluates to " + show(a) + ",\n" + eb + " evaluates to " + show(b) + ", and\n" + et +
                                    " evaluates to " + show(tol) + ".");
}
// 4-ULP comparison of EXPECT_DOUBLE_EQ / EXPECT_FLOAT_EQ on the biased integer representation
template <class F, class I> bool almostEqualUlps(F a, F b) {
  if (std::isnan(a) || std::isnan(b)) return false;
  I ia, ib;
  std::memcpy(&ia, &a, sizeof(F));
  std::memcpy(&ib, &b, sizeof(F));
  const I sign = I(1) << (8 * sizeof(I) - 1);
  auto biased = [sign](I v) { return (v & sign) ? I(~v + 1) : I(v | sign); };
  const I ba = biased(ia), bb = biased(ib);
  return (ba > bb ? ba - bb : bb - ba) <= 4;
}
inline AssertionResult CmpDoubleEq(const char *ea, const char *eb, double a, double b) {
  if (almostEqualUlps<double, unsigned long long>(a, b)) return AssertionSuccess();
  return AssertionResult(false, std::string("Expected equality of these values:\n  ") + ea + "\n    Which is: " + show(a) + "\n  " + eb + "\n    Which is: " + show(b));
}
inline AssertionResult CmpFloatEq(const char *ea, const char *eb, float a, float b) {
  if (almostEqualUlps<float, unsigned int>(a, b)) return AssertionSuccess();
  return AssertionResult(false, std::string("Expected equality of these values:\n  ") + ea + "\n    Which is: " + show(a) + "\n  " + eb + "\n    Which is: " + show(b));
}
template <class V, class M> AssertionResult CmpThat(const char *ev, const V &v, const M &matcher) {
  if (matcher.Matches(v)) return AssertionSuccess();
  return AssertionResult(false, std::string("Value of: ") + ev + "\nExpected: " + matcher.Describe() + "\n  Actual: " + show(v));
}
}  // namespace internal

// ---- the one gmock matcher the reference's tests use: ASSERT_THAT(x, ::testing::DoubleNear(expected, max_abs_error)) ----------------
class DoubleNearMatcher {
  double expected, tolerance;
public:
  DoubleNearMatcher(double e, double t) : expected(e), tolerance(t) {}
  bool Matches(double v) const { return std::fabs(v - expected) <= tolerance; }
  std::string Describe() const { return "is approximately " + internal::show(expected) + " (absolute error <= " + internal::show(tolerance) + ")"; }
};
inline DoubleNearMatcher DoubleNear(double expected, double maxAbsError) { return DoubleNearMatcher(expected, maxAbsError); }
inline DoubleNearMatcher FloatNear(float expected, float maxAbsError) { return DoubleNearMatcher(expected, maxAbsError); }

// the base of a fixture: TEST_F(Fixture, name) runs SetUp(), the body as a member of a class derived from Fixture, TearDown()
class Test {
public:
  virtual ~Test() {}
protected:
  virtual void SetUp() {}
  virtual void TearDown() {}
  virtual void TestBody() = 0;
public:
  void gtlRun() {
    SetUp();
    struct AtEnd { Test *t; ~AtEnd() { t->TearDown(); } } atEnd{this};   // (also when the body throws, as GoogleTest does)
    TestBody();
  }
};

inline void InitGoogleTest(int *, char **) {}
inline void InitGoogleMock(int *, char **) {}

namespace internal {
inline bool wildcardMatch(const char *pat, const char *s) {
  if (!*pat) return !*s;
  if (*pat == '*') return wildcardMatch(pat + 1, s) || (*s && wildcardMatch(pat, s + 1));
  return *s && (*pat == '?' || *pat == *s) && wildcardMatch(pat + 1, s + 1);
}
inline bool selected(const std::string &filter, const std::string &full) {   // "pos1:pos2-neg1:neg2", as --gtest_filter
  const size_t dash = filter.find('-');
  const std::string pos = dash == std::string::npos ? filter : filter.substr(0, dash), neg = dash == std::string::npos ? "" : filter.substr(dash + 1);
  auto any = [&](const std::string &list) {
    size_t b = 0;
    while (b <= list.size()) {
      size_t e = list.find(':', b);
      if (e == std::string::npos) e = list.size();
      if (e > b && wildcardMatch(list.substr(b, e - b).c_str(), full.c_str())) return true;
      b = e + 1;
    }
    return false;
  };
  return (pos.empty() || any(pos)) && !(neg.size() && any(neg));
}
inline int runAll(int argc, char **argv) {
  std::string filter = "*";
  bool list = false;
  for (int i = 1; i < argc; ++i) {
    if (!std::strncmp(argv[i], "--gtest_filter=", 15)) filter = argv[i] + 15;
    else if (!std::strcmp(argv[i], "--gtest_list_tests")) list = true;
  }
  std::vector<const TestInfo *> todo;
  for (const auto &t : registry()) if (selected(filter, t.suite + "." + t.name)) todo.push_back(&t);
  if (list) { for (auto *t : todo) std::cout << t->suite << "." << t->name << "\n"; return 0; }
  std::cout << "[==========] Running " << todo.size() << " tests." << std::endl;
  std::vector<std::string> failed;
  const auto t0 = std::chrono::steady_clock::now();
  for (auto *t : todo) {
    const std::string full = t->suite + "." + t->name;
    std::cout << "[ RUN      ] " << full << std::endl;
    state().failuresInCurrentTest = 0;
    const auto a = std::chrono::steady_clock::now();
    try { t->body(); }
    catch (const std::exception &e) { ++state().failuresInCurrentTest; std::cout << "unknown file: Failure\nC++ exception with description \"" << e.what() << "\" thrown in the test body." << std::endl; }
    catch (...) { ++state().failuresInCurrentTest; std::cout << "unknown file: Failure\nUnknown C++ exception thrown in the test body." << std::endl; }
    const long ms = (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - a).count();
    if (state().failuresInCurrentTest) { failed.push_back(full); std::cout << "[  FAILED  ] " << full << " (" << ms << " ms)" << std::endl; }
    else std::cout << "[       OK ] " << full << " (" << ms << " ms)" << std::endl;
  }
  const long total = (long)std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0).count();
  std::cout << "[==========] " << todo.size() << " tests ran. (" << total << " ms total)" << std::endl;
  std::cout << "[  PASSED  ] " << todo.size() - failed.size() << " tests." << std::endl;
  if (!failed.empty()) {
    std::cout << "[  FAILED  ] " << failed.size() << " tests, listed below:" << std::endl;
    for (auto &f : failed) std::cout << "[  FAILED  ] " << f << std::endl;
  }
  return failed.empty() ? 0 : 1;
}
}  // namespace internal
}  // namespace testing

#define RUN_ALL_TESTS() ::testing::internal::runAll(gtlArgc(), gtlArgv())
namespace testing { namespace internal {
inline int &argcSlot() { static int c = 0; return c; }
inline char **&argvSlot() { static char **v = nullptr; return v; }
} }
inline int gtlArgc() { return ::testing::internal::argcSlot(); }
inline char **gtlArgv() { return ::testing::internal::argvSlot(); }

#define GTL_NAME_(suite, name) suite##_##name##_gtl_test
#define TEST_F(fixture, name)                                                                                 \
  class GTL_NAME_(fixture, name) : public fixture { void TestBody() override; };                              \
  static void GTL_NAME_(fixture, name##_run)() { GTL_NAME_(fixture, name) t; t.gtlRun(); }                     \
  static ::testing::internal::Registrar GTL_NAME_(fixture, name##_registrar)(#fixture, #name, &GTL_NAME_(fixture, name##_run)); \
  void GTL_NAME_(fixture, name)::TestBody()
#define TEST(suite, name)                                                                                     \
  static void GTL_NAME_(suite, name)();                                                                       \
  static ::testing::internal::Registrar GTL_NAME_(suite, name##_registrar)(#suite, #name, &GTL_NAME_(suite, name)); \
  static void GTL_NAME_(suite, name)()

// A failed check evaluates to `AssertHelper = Message() << ...`; the fatal forms put a `return` in front (test bodies return void).
#define GTL_CHECK_(result_expr, on_failure)                                                                   \
  switch (0) case 0: default:                                                                                 \
    if (const ::testing::AssertionResult gtl_ar = (result_expr)) ;                                            \
    else on_failure ::testing::internal::AssertHelper(__FILE__, __LINE__, gtl_ar.message()) = ::testing::Message()
#define GTL_NONFATAL_
#define GTL_FATAL_ return

#define EXPECT_TRUE(c) GTL_CHECK_(::testing::internal::CmpBool(#c, static_cast<bool>(c), true), GTL_NONFATAL_)
#define EXPECT_FALSE(c) GTL_CHECK_(::testing::internal::CmpBool(#c, static_cast<bool>(c), false), GTL_NONFATAL_)
#define ASSERT_TRUE(c) GTL_CHECK_(::testing::internal::CmpBool(#c, static_cast<bool>(c), true), GTL_FATAL_)
#define ASSERT_FALSE(c) GTL_CHECK_(::testing::internal::CmpBool(#c, static_cast<bool>(c), false), GTL_FATAL_)
#define EXPECT_EQ(a, b) GTL_CHECK_(::testing::internal::CmpEQ(#a, #b, a, b), GTL_NONFATAL_)
#define EXPECT_NE(a, b) GTL_CHECK_(::testing::internal::CmpNE(#a, #b, a, b), GTL_NONFATAL_)
#define EXPECT_LT(a, b) GTL_CHECK_(::testing::internal::CmpLT(#a, #b, a, b), GTL_NONFATAL_)
#define EXPECT_LE(a, b) GTL_CHECK_(::testing::internal::CmpLE(#a, #b, a, b), GTL_NONFATAL_)
#define EXPECT_GT(a, b) GTL_CHECK_(::testing::internal::CmpGT(#a, #b, a, b), GTL_NONFATAL_)
#define EXPECT_GE(a, b) GTL_CHECK_(::testing::internal::CmpGE(#a, #b, a, b), GTL_NONFATAL_)
#define ASSERT_EQ(a, b) GTL_CHECK_(::testing::internal::CmpEQ(#a, #b, a, b), GTL_FATAL_)
#define ASSERT_NE(a, b) GTL_CHECK_(::testing::internal::CmpNE(#a, #b, a, b), GTL_FATAL_)
#define ASSERT_LT(a, b) GTL_CHECK_(::testing::internal::CmpLT(#a, #b, a, b), GTL_FATAL_)
#define ASSERT_LE(a, b) GTL_CHECK_(::testing::internal::CmpLE(#a, #b, a, b), GTL_FATAL_)
#define ASSERT_GT(a, b) GTL_CHECK_(::testing::internal::CmpGT(#a, #b, a, b), GTL_FATAL_)
#define ASSERT_GE(a, b) GTL_CHECK_(::testing::internal::CmpGE(#a, #b, a, b), GTL_FATAL_)
#define EXPECT_NEAR(a, b, tol) GTL_CHECK_(::testing::internal::CmpNear(#a, #b, #tol, a, b, tol), GTL_NONFATAL_)
#define ASSERT_NEAR(a, b, tol) GTL_CHECK_(::testing::internal::CmpNear(#a, #b, #tol, a, b, tol), GTL_FATAL_)
#define EXPECT_DOUBLE_EQ(a, b) GTL_CHECK_(::testing::internal::CmpDoubleEq(#a, #b, a, b), GTL_NONFATAL_)
#define ASSERT_DOUBLE_EQ(a, b) GTL_CHECK_(::testing::internal::CmpDoubleEq(#a, #b, a, b), GTL_FATAL_)
#define EXPECT_FLOAT_EQ(a, b) GTL_CHECK_(::testing::internal::CmpFloatEq(#a, #b, a, b), GTL_NONFATAL_)
#define ASSERT_FLOAT_EQ(a, b) GTL_CHECK_(::testing::internal::CmpFloatEq(#a, #b, a, b), GTL_FATAL_)
#define EXPECT_THAT(v, m) GTL_CHECK_(::testing::internal::CmpThat(#v, v, m), GTL_NONFATAL_)
#define ASSERT_THAT(v, m) GTL_CHECK_(::testing::internal::CmpThat(#v, v, m), GTL_FATAL_)
#define GTL_THROWS_(stmt, extype, want)                                                                      \
  [&]() -> ::testing::AssertionResult {                                                                       \
    bool threw = false;                                                                                       \
    try { stmt; } catch (const extype &) { threw = true; } catch (...) { return ::testing::AssertionResult(false, #stmt " throws another type than " #extype); } \
    return threw == want ? ::testing::AssertionSuccess() : ::testing::AssertionResult(false, want ? #stmt " does not throw " #extype : #stmt " throws " #extype); \
  }()
#define EXPECT_THROW(stmt, extype) GTL_CHECK_(GTL_THROWS_(stmt, extype, true), GTL_NONFATAL_)
#define ASSERT_THROW(stmt, extype) GTL_CHECK_(GTL_THROWS_(stmt, extype, true), GTL_FATAL_)
#define GTL_NOTHROW_(stmt)                                                                                   \
  [&]() -> ::testing::AssertionResult {                                                                       \
    try { stmt; } catch (const std::exception &e) { return ::testing::AssertionResult(false, std::string(#stmt " throws: ") + e.what()); } \
    catch (...) { return ::testing::AssertionResult(false, #stmt " throws"); }                                \
    return ::testing::AssertionSuccess();                                                                     \
  }()
#define EXPECT_NO_THROW(stmt) GTL_CHECK_(GTL_NOTHROW_(stmt), GTL_NONFATAL_)
#define ASSERT_NO_THROW(stmt) GTL_CHECK_(GTL_NOTHROW_(stmt), GTL_FATAL_)
#define ADD_FAILURE() ::testing::internal::AssertHelper(__FILE__, __LINE__, "Failed") = ::testing::Message()
#define FAIL() return ::testing::internal::AssertHelper(__FILE__, __LINE__, "Failed") = ::testing::Message()
#define SUCCEED() (void)0

#ifndef GTEST_LITE_NO_MAIN
int main(int argc, char **argv) {
  ::testing::internal::argcSlot() = argc;
  ::testing::internal::argvSlot() = argv;
  return RUN_ALL_TESTS();
}
#endif
#endif
