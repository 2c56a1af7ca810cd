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
// the names of namespace open_spiel, one by one (a using-directive for hip would make `algorithms::X` ambiguous
// between open_spiel::algorithms and open_spiel::hip::algorithms)
using hip::Action; using hip::Player; using hip::ActionsAndProbs;
using hip::kInvalidAction; using hip::kChancePlayerId; using hip::kTerminalPlayerId; using hip::kDefaultPlayerId;
using hip::kSimultaneousPlayerId; using hip::kInvalidPlayer; using hip::kMeanFieldPlayerId;
using hip::Game; using hip::State; using hip::BatchedState; using hip::LoadGame; using hip::LoadGameAsTurnBased;
using hip::RegisteredGames; using hip::RegisteredNames; using hip::RegisteredGameTypes;
using hip::SpielFatalError; using hip::SpielException;
using hip::TensorLayout; using hip::StateType; using hip::GameType; using hip::GameParameter; using hip::GameParameters; using hip::GameParametersFromString;
using hip::GameParametersToString;
using hip::SerializeGameAndState; using hip::DeserializeGameAndState;
using hip::Policy; using hip::TabularPolicy; using hip::UniformPolicy; using hip::PreferredActionPolicy;
using hip::GetUniformPolicy; using hip::GetFirstActionPolicy; using hip::GetEmptyTabularPolicy; using hip::ToTabularPolicy;
using hip::GetPrefActionPolicy;
using hip::Bot; using hip::EvaluateBots; using hip::SampleAction;
using hip::MakeUniformRandomBot; using hip::MakeStatefulRandomBot; using hip::MakePolicyBot; using hip::MakeFixedActionPreferenceBot;
using hip::Observer; using hip::Observation; using hip::IIGObservationType; using hip::PrivateInfoType;
using hip::kDefaultObsType; using hip::kInfoStateObsType; using hip::SpanTensor; using hip::SpanTensorInfo;
using hip::Near; using hip::UniformProbabilitySampler; using hip::operator<<;
namespace algorithms { using namespace hip::algorithms; }
namespace kuhn_poker { using namespace hip::kuhn_poker; }
namespace leduc_poker { using namespace hip::leduc_poker; }
namespace efg_game { using namespace hip::efg_game; }
namespace tic_tac_toe { using namespace hip::tic_tac_toe; }
namespace connect_four { using namespace hip::connect_four; }
// the struct API (spiel.h:235-299) and utils/status.h
using hip::SpielStruct; using hip::StateStruct; using hip::ObservationStruct; using hip::ActionStruct; using hip::GameParametersStruct;
using hip::SafeActionCast; using hip::LoadGameFromJson; using hip::Status; using hip::StatusValue; using hip::OkStatus; using hip::ErrorStatus;
// spiel_utils.h:119-137: the default fatal-error handler prints and exits
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
// spiel_utils.h:295-322: the check that names game and state, and the debug checks (enabled, as in the reference's
// default build)
#define SPIEL_CHECK_TRUE_WSI(x, e, g, s)                                                                  \
  do { if (!(x)) open_spiel::SpielFatalErrorShim(std::string(__FILE__) + ":" + std::to_string(__LINE__) + " CHECK_TRUE(" #x ") " + (e) + " game: " + (g).ToString() + " state: " + (s).ToString()); } while (0)
#define SPIEL_DCHECK_GE(x, y) SPIEL_CHECK_GE(x, y)
#define SPIEL_DCHECK_GT(x, y) SPIEL_CHECK_GT(x, y)
#define SPIEL_DCHECK_LE(x, y) SPIEL_CHECK_LE(x, y)
#define SPIEL_DCHECK_LT(x, y) SPIEL_CHECK_LT(x, y)
#define SPIEL_DCHECK_EQ(x, y) SPIEL_CHECK_EQ(x, y)
#define SPIEL_DCHECK_NE(x, y) SPIEL_CHECK_NE(x, y)
#define SPIEL_DCHECK_TRUE(x) SPIEL_CHECK_TRUE(x)
#define SPIEL_DCHECK_FALSE(x) SPIEL_CHECK_FALSE(x)
#define SPIEL_DCHECK_FLOAT_EQ(x, y) SPIEL_CHECK_FLOAT_EQ(x, y)
#define SPIEL_DCHECK_FLOAT_NEAR(x, y, e) SPIEL_CHECK_FLOAT_NEAR(x, y, e)
#endif  // OSG_INCLUDE_OPEN_SPIEL_SPIEL_H_
