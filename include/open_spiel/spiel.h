// Source-level drop-in for the hot path: `#include "open_spiel/spiel.h"` (and the other reference header names in
// this directory, which all forward here) with -I <repo>/include in front of — or instead of — the reference's
// include path resolves to the MI355X host mirror (open_spiel_amd/csrc/host/osg_spiel.h over the C-ABI,
// libosg_hip.so), with its names opened in the namespaces the reference declares them in: open_spiel::Game /
// State / LoadGame / Policy / Bot ... (open_spiel/spiel.h:301-1314, policy.h, spiel_bots.h) and
// open_spiel::algorithms::{MCTSBot, CFRSolver, ExternalSamplingMCCFRSolver, ...}.  A program written against the
// reference for this path compiles unchanged (tests/dropin/user_program.cc does, and so do the reference's own
// unit-test sources: tests/native/Makefile.reftests), for the five games of the path.
#ifndef OSG_INCLUDE_OPEN_SPIEL_SPIEL_H_
#define OSG_INCLUDE_OPEN_SPIEL_SPIEL_H_
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <sstream>

#include "../../open_spiel_amd/csrc/host/osg_spiel.h"

namespace open_spiel {
// the names of namespace open_spiel the test sources use, one by one (a using-directive for hip would make
// `algorithms::X` ambiguous between open_spiel::algorithms and open_spiel::hip::algorithms)
using hip::Action; using hip::Player; using hip::ActionsAndProbs;
using hip::kInvalidAction; using hip::kChancePlayerId; using hip::kTerminalPlayerId;
using hip::Game; using hip::State; using hip::BatchedState; using hip::LoadGame; using hip::SpielFatalError;
using hip::SerializeGameAndState; using hip::DeserializeGameAndState;
using hip::Policy; using hip::TabularPolicy; using hip::UniformPolicy; using hip::PreferredActionPolicy;
using hip::GetUniformPolicy; using hip::GetFirstActionPolicy; using hip::GetEmptyTabularPolicy; using hip::ToTabularPolicy;
using hip::Bot; using hip::EvaluateBots; using hip::SampleAction;
namespace algorithms { using namespace hip::algorithms; }
namespace kuhn_poker { using namespace hip::algorithms::kuhn_poker; }
// spiel_utils.h:140-250: the checks the tests use
[[noreturn]] inline void SpielFatalErrorShim(const std::string& msg) { std::cerr << "Spiel Fatal Error: " << msg << std::endl; std::exit(1); }
}  // namespace open_spiel
#define OSG_SHIM_CHECK_OP(x_exp, op, y_exp)                                                              \
  do {                                                                                                   \
    auto x = x_exp; auto y = y_exp;                                                                      \
    if (!((x)op(y))) {                                                                                   \
      std::ostringstream o; o << __FILE__ << ":" << __LINE__ << " " #x_exp " " #op " " #y_exp " failed (" << x << " vs " << y << ")"; \
      open_spiel::SpielFatalErrorShim(o.str());                                                          \
    }                                                                                                    \
  } while (0)
#define SPIEL_CHECK_GE(x, y) OSG_SHIM_CHECK_OP(x, >=, y)
#define SPIEL_CHECK_GT(x, y) OSG_SHIM_CHECK_OP(x, >, y)
#define SPIEL_CHECK_LE(x, y) OSG_SHIM_CHECK_OP(x, <=, y)
#define SPIEL_CHECK_LT(x, y) OSG_SHIM_CHECK_OP(x, <, y)
#define SPIEL_CHECK_EQ(x, y) OSG_SHIM_CHECK_OP(x, ==, y)
#define SPIEL_CHECK_NE(x, y) OSG_SHIM_CHECK_OP(x, !=, y)
#define SPIEL_CHECK_TRUE(x) do { if (!(x)) open_spiel::SpielFatalErrorShim(std::string(__FILE__) + ":" + std::to_string(__LINE__) + " CHECK_TRUE(" #x ")"); } while (0)
#define SPIEL_CHECK_FALSE(x) do { if (x) open_spiel::SpielFatalErrorShim(std::string(__FILE__) + ":" + std::to_string(__LINE__) + " CHECK_FALSE(" #x ")"); } while (0)
#define SPIEL_CHECK_FLOAT_NEAR(x_exp, y_exp, eps)                                                        \
  do {                                                                                                   \
    auto x = x_exp; auto y = y_exp;                                                                      \
    if (!(std::fabs(x - y) <= (eps))) {                                                                  \
      std::ostringstream o; o << __FILE__ << ":" << __LINE__ << " " #x_exp " ~ " #y_exp " failed (" << x << " vs " << y << ")"; \
      open_spiel::SpielFatalErrorShim(o.str());                                                          \
    }                                                                                                    \
  } while (0)
#define SPIEL_CHECK_FLOAT_EQ(x, y) SPIEL_CHECK_FLOAT_NEAR(x, y, 1e-5)
#endif  // OSG_INCLUDE_OPEN_SPIEL_SPIEL_H_
