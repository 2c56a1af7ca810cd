// Host-side C++ mirror of the reference's API for the hot path, over the C-ABI.
//
// Same class / method names and argument meaning as the reference so that code
// written against open_spiel::{Game, State, algorithms::MCTSBot, CFRSolver, ...}
// reads the same here (namespace open_spiel::hip):
//
//   Game / State / LoadGame         open_spiel/spiel.h:301-916, 927-1255, 1314
//   Evaluator / RandomRolloutEvaluator / SearchNode / MCTSBot
//                                   open_spiel/algorithms/mcts.h:83-220
//   CFRInfoStateValues / CFRSolverBase / CFRSolver / CFRPlusSolver
//                                   open_spiel/algorithms/cfr.h:42-357
//   ExternalSamplingMCCFRSolver     open_spiel/algorithms/external_sampling_mccfr.h:57-113
//
// plus BatchedState, the batch form the device actually wants.  Header-only;
// every rule evaluation happens in libosg_hip.so (HIP, gfx950).  Errors become
// SpielException (the reference's pybind handler does the same,
// python/pybind11/pyspiel.cc:831-837).  Objects are single-threaded like the
// reference's State / Bot / solver objects.
#ifndef OSG_HOST_OSG_SPIEL_H_
#define OSG_HOST_OSG_SPIEL_H_

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <iomanip>
#include <map>
#include <optional>
#include <sstream>
#include <limits>
#include <memory>
#include <random>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../../include/osg_abi.h"
#include "osg_json.h"

namespace open_spiel {
constexpr const char* kSerializeStartingState = "starting_state=";  // spiel.h:50
namespace hip {

using Action = int64_t;  // spiel_utils.h:134-135
using Player = int;
constexpr Player kChancePlayerId = OSG_CHANCE_PLAYER;      // spiel_globals.h:26-56
constexpr Player kTerminalPlayerId = OSG_TERMINAL_PLAYER;
constexpr Action kInvalidAction = OSG_INVALID_ACTION;      // spiel_globals.h:82
constexpr Player kDefaultPlayerId = 0, kSimultaneousPlayerId = -2, kInvalidPlayer = -3, kMeanFieldPlayerId = -5;  // :26-56
using ActionsAndProbs = std::vector<std::pair<Action, double>>;  // spiel.h:224

struct SpielException : public std::runtime_error {
  using std::runtime_error::runtime_error;
};
[[noreturn]] inline void SpielFatalError(const std::string& msg) { throw SpielException(msg); }
inline void Check(int rc) {
  if (rc != OSG_OK) SpielFatalError(std::string("osg error ") + std::to_string(rc) + ": " + osg_last_error());
}

// spiel_utils.h:185-200
template <typename T>
bool Near(T a, T b, T epsilon) {
  static_assert(std::is_floating_point<T>::value, "Near() is only for floating point args.");
  return std::fabs(a - b) <= epsilon;
}
template <typename T>
bool Near(T a, T b) {
  static_assert(std::is_floating_point<T>::value, "Near() is only for floating point args.");
  return Near(a, b, static_cast<T>(std::numeric_limits<float>::epsilon() * 128));
}
// spiel_utils.h:44-100: ostream operators for the containers the checks print
template <typename T> std::ostream& operator<<(std::ostream& stream, const std::vector<T>& v);
template <typename T, typename U> std::ostream& operator<<(std::ostream& stream, const std::pair<T, U>& v) {
  return stream << "(" << v.first << "," << v.second << ")";
}
template <typename T> std::ostream& operator<<(std::ostream& stream, const std::vector<T>& v) {
  stream << "[";
  for (const auto& element : v) stream << element << " ";
  return stream << "]";
}
template <typename T> std::ostream& operator<<(std::ostream& stream, const std::optional<T>& v) {
  return v.has_value() ? stream << *v : stream << "nullopt";
}
template <typename T> std::ostream& operator<<(std::ostream& stream, const std::unique_ptr<T>& v) { return stream << *v; }
// spiel.h:1303-1311: the sampler ResampleFromInfostate takes
class UniformProbabilitySampler {
 public:
  UniformProbabilitySampler(int seed, double min = 0., double max = 1.) : seed_(seed), rng_(seed_), dist_(min, max) {}
  UniformProbabilitySampler(double min = 0., double max = 1.) : rng_(seed_), dist_(min, max) {}
  double operator()() { return dist_(rng_); }

 private:
  int seed_ = 0;  // (the reference seeds from the clock: callers that care pass a seed)
  std::mt19937 rng_;
  std::uniform_real_distribution<double> dist_;
};

// One engine context per (process, device), created on first use with its own stream.
class Context {
 public:
  static osg_ctx* Default(int device = 0) {
    static std::unordered_map<int, std::unique_ptr<Context>> table;
    auto it = table.find(device);
    if (it == table.end()) it = table.emplace(device, std::unique_ptr<Context>(new Context(device))).first;
    return it->second->ctx_;
  }
  ~Context() { osg_ctx_destroy(ctx_); }

 private:
  explicit Context(int device) { Check(osg_ctx_create(device, nullptr, /*own_stream=*/1, &ctx_)); }
  osg_ctx* ctx_ = nullptr;
};

// Contiguous slice [first, first + count) of `total` units owned by `rank`: the first total % world ranks
// own one extra unit (open_spiel_amd/distributed.py shard_range).  Every device RNG stream is keyed by
// the GLOBAL unit index, so results do not depend on the world size.
inline std::pair<int64_t, int64_t> ShardRange(int64_t total, int rank, int world) {
  if (world < 1 || rank < 0 || rank >= world) SpielFatalError("ShardRange: bad rank / world");
  const int64_t base = total / world, extra = total % world;
  return {rank * base + std::min<int64_t>(rank, extra), base + (rank < extra ? 1 : 0)};
}

// One RCCL communicator per (process, GPU) for C++ hosts (the Python path uses torch.distributed, which
// is RCCL as well).  The reference has no distributed runtime; this is the exchange step of SURVEY.md 8e.
// Rank 0 calls NewId() and hands the 128 bytes to the other ranks out of band.
class Communicator {
 public:
  using Id = std::array<char, OSG_COMM_ID_BYTES>;
  static Id NewId() {
    Id id{};
    Check(osg_comm_unique_id(id.data()));
    return id;
  }
  Communicator(int rank, int world, const Id& id, int device = 0) {
    Check(osg_comm_create(Context::Default(device), rank, world, id.data(), &c_));
  }
  // The one-shot kind for the path's latency-bound messages (osg_comm_oneshot_*: every rank's window mapped by
  // every peer through hipIpc, one launch per all-reduce, sums in rank order — bit-identical on all ranks; no RCCL):
  // every rank makes one, hands LocalHandle() to the others out of band (the channel the Id would travel on) and
  // calls Connect() with all of them in rank order.  AllReduceSum / Begin / End as with the RCCL kind.
  using Handle = std::array<char, OSG_ONESHOT_HANDLE_BYTES>;
  struct OneShotTag {};
  Communicator(OneShotTag, int rank, int world, int64_t max_doubles, int device = 0) {
    Check(osg_comm_oneshot_create(Context::Default(device), rank, world, max_doubles, &c_));
  }
  Handle LocalHandle() const {
    Handle h{};
    Check(osg_comm_oneshot_handle(c_, h.data()));
    return h;
  }
  void Connect(const std::vector<Handle>& handles_in_rank_order) {
    std::vector<char> flat;
    for (const Handle& h : handles_in_rank_order) flat.insert(flat.end(), h.begin(), h.end());
    if (static_cast<int>(handles_in_rank_order.size()) != world()) SpielFatalError("Communicator::Connect: one handle per rank");
    Check(osg_comm_oneshot_connect(c_, flat.data()));
  }
  ~Communicator() { osg_comm_destroy(c_); }
  Communicator(const Communicator&) = delete;
  Communicator& operator=(const Communicator&) = delete;
  int rank() const { return osg_comm_rank(c_); }
  int world() const { return osg_comm_world(c_); }
  std::pair<int64_t, int64_t> Shard(int64_t total) const { return ShardRange(total, rank(), world()); }
  // In place, on the context's stream (ordered with the kernels before and after it).
  void AllReduceSum(double* d_buf, int64_t n) { Check(osg_allreduce_sum_f64(c_, d_buf, n)); }
  void AllReduceSum(int32_t* d_buf, int64_t n) { Check(osg_allreduce_sum_i32(c_, d_buf, n)); }
  // The asynchronous form: the collective runs on the communicator's own stream; kernels issued on the
  // context's stream between Begin and End overlap it, End orders the context's stream after it (no host wait).
  void BeginAllReduceSum(double* d_buf, int64_t n) { Check(osg_allreduce_sum_f64_begin(c_, d_buf, n)); }
  void EndAllReduce() { Check(osg_allreduce_end(c_)); }
  // Waits for the collectives issued so far and raises if a one-shot call timed out (its buffer then holds NaN in
  // the chunks that were not reduced): call before trusting the result of the last collective of a job.
  void CheckHealth() { Check(osg_comm_check(c_)); }

 private:
  osg_comm* c_ = nullptr;
};

// ---- game_parameters.h:37-180: GameParameter / GameParameters, for the parameters the path's games take ----
class GameParameter;
using GameParameters = std::map<std::string, GameParameter>;
class GameParameter {
 public:
  enum class Type { kUnset = -1, kInt, kDouble, kString, kBool, kGame };
  explicit GameParameter(Type type = Type::kUnset, bool is_mandatory = false) : is_mandatory_(is_mandatory), type_(type) {}
  explicit GameParameter(int value, bool is_mandatory = false) : is_mandatory_(is_mandatory), int_value_(value), type_(Type::kInt) {}
  explicit GameParameter(double value, bool is_mandatory = false)
      : is_mandatory_(is_mandatory), double_value_(value), type_(Type::kDouble) {}
  explicit GameParameter(std::string value, bool is_mandatory = false)
      : is_mandatory_(is_mandatory), string_value_(std::move(value)), type_(Type::kString) {}
  explicit GameParameter(const char* value, bool is_mandatory = false)
      : is_mandatory_(is_mandatory), string_value_(value), type_(Type::kString) {}
  explicit GameParameter(bool value, bool is_mandatory = false) : is_mandatory_(is_mandatory), bool_value_(value), type_(Type::kBool) {}
  bool has_int_value() const { return type_ == Type::kInt; }
  bool has_double_value() const { return type_ == Type::kDouble; }
  bool has_string_value() const { return type_ == Type::kString; }
  bool has_bool_value() const { return type_ == Type::kBool; }
  bool has_game_value() const { return type_ == Type::kGame; }
  Type type() const { return type_; }
  bool is_mandatory() const { return is_mandatory_; }
  int int_value() const { Want(Type::kInt); return int_value_; }
  double double_value() const { Want(Type::kDouble); return double_value_; }
  const std::string& string_value() const { Want(Type::kString); return string_value_; }
  bool bool_value() const { Want(Type::kBool); return bool_value_; }
  std::string ToString() const {  // game_parameters.cc:48-66: the text a game string carries
    switch (type_) {
      case Type::kInt: return std::to_string(int_value_);
      case Type::kDouble: { std::ostringstream o; o << double_value_; return o.str(); }
      case Type::kString: return string_value_;
      case Type::kBool: return bool_value_ ? "True" : "False";
      default: return "";
    }
  }

 private:
  void Want(Type t) const { if (type_ != t) SpielFatalError("GameParameter: wrong value type requested"); }
  bool is_mandatory_ = false;
  int int_value_ = 0;
  double double_value_ = 0;
  std::string string_value_;
  bool bool_value_ = false;
  Type type_ = Type::kUnset;
};
// game_parameters.cc:120-165 / :167-205: "name(k=v,k=v)" <-> {"name": ..., k: v} (values of a parsed string stay strings)
inline GameParameters GameParametersFromString(const std::string& game_string) {
  GameParameters params;
  const size_t open = game_string.find('(');
  params["name"] = GameParameter(game_string.substr(0, open));
  if (open == std::string::npos) return params;
  const size_t close = game_string.rfind(')');
  if (close == std::string::npos || close < open) SpielFatalError("GameParametersFromString: missing ')' in " + game_string);
  const std::string body = game_string.substr(open + 1, close - open - 1);
  size_t pos = 0;
  while (pos < body.size()) {
    size_t comma = body.find(',', pos);
    if (comma == std::string::npos) comma = body.size();
    const std::string item = body.substr(pos, comma - pos);
    pos = comma + 1;
    const size_t eq = item.find('=');
    if (eq == std::string::npos) SpielFatalError("GameParametersFromString: expected key=value in " + game_string);
    params[item.substr(0, eq)] = GameParameter(item.substr(eq + 1));
  }
  return params;
}
inline std::string GameParametersToString(const GameParameters& params) {
  auto name = params.find("name");
  if (name == params.end()) SpielFatalError("GameParametersToString: no 'name' parameter");
  std::string str = name->second.string_value(), args;
  for (const auto& kv : params) {
    if (kv.first == "name") continue;
    args += (args.empty() ? "" : ",") + kv.first + "=" + kv.second.ToString();
  }
  return args.empty() ? str : str + "(" + args + ")";
}

// spiel.h:55-175: GameType, as the five games of the path register it
struct GameType {
  std::string short_name, long_name;
  enum class Dynamics { kSimultaneous, kSequential, kMeanField };
  Dynamics dynamics = Dynamics::kSequential;
  enum class ChanceMode { kDeterministic, kExplicitStochastic, kSampledStochastic };
  ChanceMode chance_mode = ChanceMode::kDeterministic;
  enum class Information { kOneShot, kPerfectInformation, kImperfectInformation };
  Information information = Information::kPerfectInformation;
  enum class Utility { kZeroSum, kConstantSum, kGeneralSum, kIdentical };
  Utility utility = Utility::kZeroSum;
  enum class RewardModel { kRewards, kTerminal };
  RewardModel reward_model = RewardModel::kTerminal;
  int max_num_players = 2, min_num_players = 2;
  bool provides_information_state_string = true, provides_information_state_tensor = false;
  bool provides_observation_string = true, provides_observation_tensor = true;
  GameParameters parameter_specification;
  bool default_loadable = true;
  bool provides_factored_observation_string = false;
  bool is_concrete = true;
  bool provides_information_state() const { return provides_information_state_tensor || provides_information_state_string; }
  bool provides_observation() const { return provides_observation_tensor || provides_observation_string; }
};

class State;
class BatchedState;
struct StateStruct;
struct GameParametersStruct;
enum class TensorLayout { kHWC, kCHW };  // spiel.h:231
enum class StateType { kTerminal, kChance, kDecision, kMeanField };  // spiel_globals.h:84-92
class Policy;
class TabularPolicy;
class Observer;
struct IIGObservationType;

class Game : public std::enable_shared_from_this<Game> {
 public:
  explicit Game(const std::string& game_string, int device = 0) : string_(game_string), device_(device) {
    Check(osg_game_describe(game_string.c_str(), &desc_));
  }
  int NumDistinctActions() const { return desc_.num_distinct_actions; }
  int MaxChanceOutcomes() const { return desc_.max_chance_outcomes; }
  int NumPlayers() const { return desc_.num_players; }
  double MinUtility() const { return desc_.min_utility; }
  double MaxUtility() const { return desc_.max_utility; }
  int MaxGameLength() const { return desc_.max_game_length; }
  int MaxChanceNodesInHistory() const { return desc_.max_chance_nodes; }
  std::vector<int> ObservationTensorShape() const {
    return std::vector<int>(desc_.obs_shape, desc_.obs_shape + desc_.obs_rank);
  }
  std::vector<int> InformationStateTensorShape() const {
    return std::vector<int>(desc_.info_shape, desc_.info_shape + desc_.info_rank);
  }
  int ObservationTensorSize() const { return desc_.obs_size; }
  int InformationStateTensorSize() const { return desc_.info_size; }
  std::string ToString() const { return desc_.canonical; }
  // The registered GameType (tic_tac_toe.cc:32-49, connect_four.cc:41-58, hex.cc:37-58, kuhn_poker.cc:36-56,
  // leduc_poker.cc:45-70): every game of the path is sequential, zero-sum, with terminal rewards.
  GameType GetType() const {
    const std::string text = ToString();
    GameType t;
    t.short_name = text.substr(0, text.find('('));
    const bool poker = t.short_name == "kuhn_poker" || t.short_name == "leduc_poker";
    t.long_name = t.short_name == "tic_tac_toe" ? "Tic Tac Toe" : t.short_name == "connect_four" ? "Connect Four"
                  : t.short_name == "hex" ? "Hex" : t.short_name == "kuhn_poker" ? "Kuhn Poker" : "Leduc Poker";
    t.chance_mode = poker ? GameType::ChanceMode::kExplicitStochastic : GameType::ChanceMode::kDeterministic;
    t.information = poker ? GameType::Information::kImperfectInformation : GameType::Information::kPerfectInformation;
    t.max_num_players = poker ? 10 : 2;
    t.provides_information_state_tensor = poker;
    return t;
  }
  std::optional<double> UtilitySum() const { return 0.0; }               // spiel.h:1012-1016: zero-sum games
  int MaxMoveNumber() const { return MaxGameLength() + MaxChanceNodesInHistory(); }  // spiel.h:1125-1127
  int MaxHistoryLength() const { return MaxGameLength() + MaxChanceNodesInHistory(); }  // spiel.h:1100-1102 (sequential games)
  std::vector<int> PolicyTensorShape() const { return {NumDistinctActions()}; }  // spiel.h:1059-1061
  TensorLayout ObservationTensorLayout() const { return TensorLayout::kCHW; }   // spiel.h:1043-1045
  TensorLayout InformationStateTensorLayout() const { return TensorLayout::kCHW; }
  GameParameters GetParameters() const { return GameParametersFromString(ToString()); }
  std::string Serialize() const { return ToString(); }  // spiel.cc:793-800 (no sampled-stochastic games here)
  const std::string& GameString() const { return string_; }
  const osg_game_desc& Desc() const { return desc_; }
  osg_ctx* Ctx() const { return Context::Default(device_); }
  inline std::unique_ptr<State> NewInitialState() const;
  // spiel.h:967-971, tic_tac_toe.h:140-149, connect_four.h:182-192: a state from its struct / its JSON (the two board
  // games whose reference State has a constructor from a board)
  inline std::unique_ptr<State> NewInitialState(const StateStruct& state_struct) const;
  inline std::unique_ptr<State> NewInitialState(const Json& json) const;
  inline std::unique_ptr<State> NewInitialState(const std::string& str) const;   // spiel.h:952-954: the JSON text
  inline std::unique_ptr<State> NewInitialState(const char* str) const;          // spiel.h:957-959
  std::unique_ptr<State> NewInitialStateForPopulation(int) const {  // spiel.h:963-966: mean-field games only
    SpielFatalError("NewInitialStateForPopulation is not implemented.");
  }
  // spiel.h:1040-1054: the built-in observer for the type (nullopt = the game's default one); the only named
  // observer the path's games register is "single_tensor" (observer.cc:347-357), which IS the built-in one
  inline std::shared_ptr<Observer> MakeObserver(std::optional<IIGObservationType> iig_obs_type,
                                                const GameParameters& params = {}) const;
  inline std::unique_ptr<State> DeserializeState(const std::string& str) const;  // spiel.cc:540-580
  inline BatchedState NewInitialStates(int64_t n) const;

 private:
  std::string string_;
  int device_;
  osg_game_desc desc_{};
};

inline std::shared_ptr<const Game> LoadGame(const std::string& game_string) {  // spiel.h:1314
  return std::make_shared<const Game>(game_string);
}
inline std::shared_ptr<const Game> LoadGame(GameParameters params) {  // spiel.h:1326
  return LoadGame(GameParametersToString(params));
}
inline std::shared_ptr<const Game> LoadGame(const std::string& short_name, const GameParameters& params) {  // spiel.h:1320
  GameParameters all = params;
  all["name"] = GameParameter(short_name);
  return LoadGame(all);
}
// Entry points of the reference that lead OUTSIDE the five games of the path (game transforms, .efg files): declared
// so that code naming them compiles; calling them is an error here, as LoadGame of any other game is.
// spiel.h:1336-1341: what LoadGame can load here
inline std::vector<std::string> RegisteredNames() { return {"connect_four", "hex", "kuhn_poker", "leduc_poker", "tic_tac_toe"}; }
inline std::vector<std::string> RegisteredGames() { return RegisteredNames(); }  // spiel.h:1305 (the free function: names)
inline std::vector<GameType> RegisteredGameTypes() {                             // spiel.h:1306
  std::vector<GameType> out;
  for (const std::string& name : RegisteredNames()) out.push_back(LoadGame(name)->GetType());
  return out;
}
inline std::shared_ptr<const Game> LoadGameAsTurnBased(const std::string& name) {
  SpielFatalError("LoadGameAsTurnBased(" + name + "): game transforms are not on the MI355X path");
}
inline std::shared_ptr<const Game> LoadGameAsTurnBased(const std::string& name, const GameParameters&) {
  return LoadGameAsTurnBased(name);
}
// ---- utils/status.h:23-73 ----
enum class StatusValue { kOk = 0, kError = 1 };
class Status {
 public:
  explicit Status() : status_value_(StatusValue::kOk) {}
  Status(StatusValue status_value, const std::string& message) : status_value_(status_value), message_(message) {}
  bool ok() const { return status_value_ == StatusValue::kOk; }
  std::string message() const { return message_; }
  std::string ToString() const { return ok() ? "OK" : "ERROR: " + message_; }

 private:
  StatusValue status_value_;
  std::string message_;
};
inline Status OkStatus() { return Status(); }
inline Status ErrorStatus(const std::string& message) { return Status(StatusValue::kError, message); }
inline std::ostream& operator<<(std::ostream& os, const Status& status) { return os << status.ToString(); }

// ---- the struct API (spiel.h:235-299): structured, JSON-printable information about states, observations, actions
// and game parameters, for the two board games that define it (tic_tac_toe.h:58-76, connect_four.h:72-113) ----
struct SpielStruct {
  virtual ~SpielStruct() = default;
  std::string ToJson() const { return to_json_base().dump(); }
  virtual Json to_json_base() const = 0;
};
struct StateStruct : public SpielStruct {};
struct ObservationStruct : public SpielStruct {};
struct ActionStruct : public SpielStruct {};
struct GameParametersStruct : public SpielStruct {
  std::string game_name;  // required: which game this configures
  Json to_json_base() const override {
    Json j = Json::object();
    j["game_name"] = game_name;
    return j;
  }
};
template <typename ActionStructType>
const ActionStructType* SafeActionCast(const ActionStruct& action_struct) {  // spiel.h:291-298
  const auto* result = dynamic_cast<const ActionStructType*>(&action_struct);
  if (result == nullptr) SpielFatalError("SafeActionCast: the action struct is of another game");
  return result;
}
inline std::string DefaultPlayerString(Player player) {  // spiel_utils.cc: "Terminal", "Chance", ...
  switch (player) {
    case kTerminalPlayerId: return "Terminal";
    case kChancePlayerId: return "Chance";
    case kSimultaneousPlayerId: return "Simultaneous";
    case kInvalidPlayer: return "Invalid";
    case kMeanFieldPlayerId: return "MeanField";
    default: return "Player" + std::to_string(player);
  }
}

// games/tic_tac_toe/tic_tac_toe.h:40-76: the constants user code sizes its arrays with, and the struct types
namespace tic_tac_toe {
inline constexpr int kNumPlayers = 2;
inline constexpr int kNumRows = 3;
inline constexpr int kNumCols = 3;
inline constexpr int kNumCells = kNumRows * kNumCols;
inline constexpr int kCellStates = 1 + kNumPlayers;  // empty, 'x', 'o'
inline constexpr int kNumberStates = 5478;           // distinct reachable positions
struct TicTacToeStructContents {
  std::string current_player;
  std::vector<std::string> board;
  void from_json(const Json& j) {
    j.at("current_player").get_to(current_player);
    j.at("board").get_to(board);
  }
  Json contents_json() const {
    Json j = Json::object();
    j["board"] = Json(board);
    j["current_player"] = current_player;
    return j;
  }
};
struct TicTacToeStateStruct : public StateStruct, public TicTacToeStructContents {
  TicTacToeStateStruct() = default;
  explicit TicTacToeStateStruct(const std::string& json_str) { from_json(Json::parse(json_str)); }
  explicit TicTacToeStateStruct(const Json& j) { from_json(j); }
  Json to_json_base() const override { return contents_json(); }
};
struct TicTacToeObservationStruct : public ObservationStruct, public TicTacToeStructContents {
  TicTacToeObservationStruct() = default;
  explicit TicTacToeObservationStruct(const std::string& json_str) { from_json(Json::parse(json_str)); }
  explicit TicTacToeObservationStruct(const Json& j) { from_json(j); }
  Json to_json_base() const override { return contents_json(); }
};
struct TicTacToeActionStruct : public ActionStruct {
  int row = 0;
  int col = 0;
  TicTacToeActionStruct() = default;
  explicit TicTacToeActionStruct(const std::string& json_str) { from_json(Json::parse(json_str)); }
  explicit TicTacToeActionStruct(const Json& j) { from_json(j); }
  void from_json(const Json& j) { j.at("row").get_to(row); j.at("col").get_to(col); }
  Json to_json_base() const override {
    Json j = Json::object();
    j["col"] = col;
    j["row"] = row;
    return j;
  }
};
}  // namespace tic_tac_toe

// games/connect_four/connect_four.h:50-113
namespace connect_four {
inline constexpr int kNumPlayers = 2;
inline constexpr int kDefaultNumRows = 6;
inline constexpr int kDefaultNumCols = 7;
inline constexpr int kDefaultXInRow = 4;
inline constexpr bool kDefaultEgocentricObsTensor = false;
inline constexpr int kCellStates = 1 + kNumPlayers;  // player 0, player 1, empty
struct ConnectFourStructContents {
  std::vector<std::vector<std::string>> board;  // board[r][c], row 0 the bottom row
  std::string current_player;
  bool is_terminal = false;
  std::string winner;
  void from_json(const Json& j) {
    j.at("board").get_to(board);
    j.at("current_player").get_to(current_player);
    j.at("is_terminal").get_to(is_terminal);
    j.at("winner").get_to(winner);
  }
  Json contents_json() const {
    Json j = Json::object();
    Json rows = Json::array();
    for (const std::vector<std::string>& row : board) rows.push_back(Json(row));
    j["board"] = rows;
    j["current_player"] = current_player;
    j["is_terminal"] = is_terminal;
    j["winner"] = winner;
    return j;
  }
};
struct ConnectFourStateStruct : public StateStruct, public ConnectFourStructContents {
  ConnectFourStateStruct() = default;
  explicit ConnectFourStateStruct(const std::string& json_str) { from_json(Json::parse(json_str)); }
  explicit ConnectFourStateStruct(const Json& j) { from_json(j); }
  Json to_json_base() const override { return contents_json(); }
};
struct ConnectFourObservationStruct : public ObservationStruct, public ConnectFourStructContents {
  ConnectFourObservationStruct() = default;
  explicit ConnectFourObservationStruct(const std::string& json_str) { from_json(Json::parse(json_str)); }
  explicit ConnectFourObservationStruct(const Json& j) { from_json(j); }
  Json to_json_base() const override { return contents_json(); }
};
struct ConnectFourActionStruct : public ActionStruct {
  int column = 0;
  ConnectFourActionStruct() = default;
  explicit ConnectFourActionStruct(const std::string& json_str) { from_json(Json::parse(json_str)); }
  explicit ConnectFourActionStruct(const Json& j) { from_json(j); }
  void from_json(const Json& j) { j.at("column").get_to(column); }
  Json to_json_base() const override {
    Json j = Json::object();
    j["column"] = column;
    return j;
  }
};
struct ConnectFourGameParams : public GameParametersStruct {
  int rows = kDefaultNumRows;
  int columns = kDefaultNumCols;
  int x_in_row = kDefaultXInRow;
  bool egocentric_obs_tensor = kDefaultEgocentricObsTensor;
  ConnectFourGameParams() { game_name = "connect_four"; }
  explicit ConnectFourGameParams(const std::string& json_str) : ConnectFourGameParams() { from_json(Json::parse(json_str)); }
  explicit ConnectFourGameParams(const Json& j) : ConnectFourGameParams() { from_json(j); }
  void from_json(const Json& j) {  // (NLOHMANN_DEFINE_TYPE_INTRUSIVE: every field must be there)
    j.at("game_name").get_to(game_name);
    j.at("rows").get_to(rows);
    j.at("columns").get_to(columns);
    j.at("x_in_row").get_to(x_in_row);
    j.at("egocentric_obs_tensor").get_to(egocentric_obs_tensor);
  }
  Json to_json_base() const override {
    Json j = Json::object();
    j["game_name"] = game_name;
    j["rows"] = rows;
    j["columns"] = columns;
    j["x_in_row"] = x_in_row;
    j["egocentric_obs_tensor"] = egocentric_obs_tensor;
    return j;
  }
};
}  // namespace connect_four

// spiel.cc:326-353: a game from JSON game parameters ({"game_name": ..., <parameter>: <bool | int | double | string>})
inline std::shared_ptr<const Game> LoadGameFromJson(const std::string& json_string) {
  const Json json = Json::parse(json_string);
  if (!json.contains("game_name")) SpielFatalError("JSON game params must contain 'game_name' key.");
  const std::string game_name = json.at("game_name").get<std::string>();
  if (game_name.empty()) SpielFatalError("JSON game params 'game_name' must not be empty.");
  GameParameters params;
  params["name"] = GameParameter(game_name);
  for (const auto& kv : json.items_object()) {
    if (kv.first == "game_name") continue;
    const Json& value = kv.second;
    if (value.is_boolean()) params[kv.first] = GameParameter(value.get<bool>());
    else if (value.is_number_integer()) params[kv.first] = GameParameter(value.get<int>());
    else if (value.is_number()) params[kv.first] = GameParameter(value.get<double>());
    else if (value.is_string()) params[kv.first] = GameParameter(value.get<std::string>());
    else SpielFatalError("Unsupported JSON value type for key: " + kv.first);
  }
  return LoadGame(params);
}
inline std::shared_ptr<const Game> LoadGame(const GameParametersStruct& params_struct) {  // spiel.h:1332-1335
  return LoadGameFromJson(params_struct.ToJson());
}
namespace efg_game {
inline std::string GetKuhnPokerEFGData() { return ""; }
inline std::shared_ptr<const Game> LoadEFGGame(const std::string&) {
  SpielFatalError("LoadEFGGame: .efg games are not on the MI355X path");
}
}  // namespace efg_game

// N states of one game in HBM.  Vector-valued results are [n, ...] row-major.
class BatchedState {
 public:
  BatchedState(std::shared_ptr<const Game> game, int64_t n) : game_(std::move(game)), n_(n) {
    Check(osg_batch_create(game_->Ctx(), game_->GameString().c_str(), n, &b_));
  }
  BatchedState(const BatchedState& o) : game_(o.game_), n_(o.n_) {  // State::Clone for the whole batch
    Check(osg_batch_create(game_->Ctx(), game_->GameString().c_str(), n_, &b_));
    Check(osg_batch_copy(b_, o.b_));
  }
  BatchedState(BatchedState&& o) noexcept : game_(std::move(o.game_)), n_(o.n_), b_(o.b_) { o.b_ = nullptr; }
  BatchedState& operator=(const BatchedState&) = delete;
  BatchedState& operator=(BatchedState&& o) noexcept {
    if (this != &o) {
      if (b_) osg_batch_destroy(b_);
      game_ = std::move(o.game_); n_ = o.n_; b_ = o.b_; o.b_ = nullptr;
    }
    return *this;
  }
  ~BatchedState() { if (b_) osg_batch_destroy(b_); }

  int64_t size() const { return n_; }
  osg_batch* handle() const { return b_; }
  const std::shared_ptr<const Game>& GetGame() const { return game_; }

  std::vector<uint32_t> LegalActionsMaskBits() const {  // [n, mask_words]
    std::vector<uint32_t> m(static_cast<size_t>(n_) * game_->Desc().mask_words);
    Check(osg_legal_mask(b_, m.data(), 1));
    return m;
  }
  // ApplyAction for every state; -1 leaves a state untouched; illegal actions are fatal.
  void ApplyActions(const std::vector<int32_t>& actions) {
    if (static_cast<int64_t>(actions.size()) != n_) SpielFatalError("ApplyActions: need one action per state");
    int64_t illegal = 0;
    Check(osg_apply(b_, actions.data(), 1, &illegal));
    if (illegal) SpielFatalError(std::to_string(illegal) + " illegal action(s) applied");
  }
  std::vector<uint8_t> IsTerminal() const {
    std::vector<uint8_t> t(n_);
    Check(osg_status_query(b_, nullptr, t.data(), nullptr, 1));
    return t;
  }
  std::vector<int8_t> CurrentPlayer() const {
    std::vector<int8_t> c(n_);
    Check(osg_status_query(b_, c.data(), nullptr, nullptr, 1));
    return c;
  }
  std::vector<double> Returns() const {  // [n, P]
    std::vector<double> r(static_cast<size_t>(n_) * game_->NumPlayers());
    Check(osg_status_query(b_, nullptr, nullptr, r.data(), 1));
    return r;
  }
  std::vector<double> ChanceOutcomeProbs() const {  // [n, MaxChanceOutcomes]
    std::vector<double> p(static_cast<size_t>(n_) * std::max(game_->MaxChanceOutcomes(), 1));
    Check(osg_chance_probs(b_, p.data(), 1));
    return p;
  }
  std::vector<float> ObservationTensor(Player player) const {
    std::vector<float> o(static_cast<size_t>(n_) * game_->ObservationTensorSize());
    Check(osg_observation(b_, player, 0, o.data(), 1));
    return o;
  }
  std::vector<float> InformationStateTensor(Player player) const {
    std::vector<float> o(static_cast<size_t>(n_) * game_->InformationStateTensorSize());
    Check(osg_observation(b_, player, 1, o.data(), 1));
    return o;
  }

 private:
  std::shared_ptr<const Game> game_;
  int64_t n_;
  osg_batch* b_ = nullptr;
};

// One state: the reference's State interface on a 1-element batch.
class State {
 public:
  explicit State(std::shared_ptr<const Game> game) : batch_(std::move(game), 1) {}
  State(const State&) = default;

  // CurrentPlayer / IsTerminal / Returns of one position come from ONE device query (osg_status_query fills all three)
  // and the legal mask from one more; both are kept until the position changes (ApplyAction, UndoAction), so the
  // per-state loop of a caller — IsTerminal, CurrentPlayer, IsChanceNode, LegalActions, ApplyAction — costs three device
  // round trips instead of six.
  Player CurrentPlayer() const { return Snap().cur; }
  bool IsTerminal() const { return Snap().term != 0; }
  bool IsChanceNode() const { return CurrentPlayer() == kChancePlayerId; }
  std::vector<double> Returns() const { return Snap().returns; }
  std::vector<double> Rewards() const { return Returns(); }  // every game here is RewardModel::kTerminal
  double PlayerReturn(Player p) const { return Returns()[p]; }
  std::vector<Action> LegalActions() const {  // sorted ascending (basic_tests.cc:784-792)
    if (!snap_.have_mask) {
      snap_.bits = batch_.LegalActionsMaskBits();
      snap_.have_mask = true;
    }
    const std::vector<uint32_t>& bits = snap_.bits;
    std::vector<Action> out;
    for (size_t w = 0; w < bits.size(); ++w)
      for (int b = 0; b < 32; ++b)
        if ((bits[w] >> b) & 1u) out.push_back(static_cast<Action>(w * 32 + b));
    return out;
  }
  std::vector<Action> LegalActions(Player player) const {  // spiel.h:366-372
    if (IsTerminal() || player != CurrentPlayer()) return {};  // at a chance node: only for kChancePlayerId
    return LegalActions();
  }
  std::vector<int> LegalActionsMask() const { return LegalActionsMask(CurrentPlayer()); }
  std::vector<int> LegalActionsMask(Player player) const {  // spiel.cc:518-524
    const Game& g = *batch_.GetGame();
    const int len = player == kChancePlayerId ? g.MaxChanceOutcomes() : g.NumDistinctActions();
    std::vector<int> mask(len, 0);
    for (Action a : LegalActions(player)) mask[a] = 1;
    return mask;
  }
  bool IsSimultaneousNode() const { return false; }  // every game of the path is GameType::Dynamics::kSequential
  bool IsMeanFieldNode() const { return false; }
  bool IsPlayerNode() const { return CurrentPlayer() >= 0; }
  bool IsInitialState() const { return history_.empty(); }
  int NumDistinctActions() const { return batch_.GetGame()->NumDistinctActions(); }
  void ApplyActionWithLegalityCheck(Action a) {  // spiel.cc:453-462
    const std::vector<Action> legal = LegalActions();
    if (std::find(legal.begin(), legal.end(), a) == legal.end())
      SpielFatalError("Current player " + std::to_string(CurrentPlayer()) + " calling ApplyAction with illegal action (" +
                      std::to_string(a) + ")");
    ApplyAction(a);
  }
  void ApplyActions(const std::vector<Action>&) { SpielFatalError("ApplyActions is not implemented: no simultaneous-move game on this path"); }
  void ApplyActionsWithLegalityChecks(const std::vector<Action>& a) { ApplyActions(a); }
  std::vector<std::string> DistributionSupport() { SpielFatalError("DistributionSupport has not been implemented"); }
  void UpdateDistribution(const std::vector<double>&) { SpielFatalError("UpdateDistribution has not been implemented"); }
  // UndoAction (spiel.h:555-571): the state before its last action.  The device holds positions, not move stacks:
  // the predecessor is rebuilt from the history (the reference offers Undo only where a game implements it —
  // kuhn_poker and tic_tac_toe on this path; here every game has it, at the cost of a replay).
  void UndoAction(Player player, Action action) {
    if (history_.empty() || history_.back().second != action || history_.back().first != player)
      SpielFatalError("UndoAction: (player, action) is not the last move of this state");
    std::vector<std::pair<Player, Action>> keep(history_.begin(), history_.end() - 1);
    BatchedState fresh(batch_.GetGame(), 1);
    // a state built from a struct / JSON / board string starts at that position, not at the empty board
    if (!starting_cells_.empty())
      Check(osg_batch_set_cells(fresh.handle(), 0, starting_cells_.data(), static_cast<int>(starting_cells_.size())));
    for (const auto& pa : keep) fresh.ApplyActions({static_cast<int32_t>(pa.second)});
    batch_ = std::move(fresh);
    snap_ = Snapshot{};
    history_ = std::move(keep);
  }
  // ResampleFromInfostate (spiel.h:778-786; kuhn_poker.cc:312-327, leduc_poker.cc:706-760): a state `player`
  // cannot tell from this one — the other players' private deals (the first NumPlayers chance outcomes, one per
  // player, in the two poker games) are drawn again, uniformly among the cards that stay consistent with
  // everything `player` has seen (its own card, the public chance outcomes), every other action is replayed.
  std::unique_ptr<State> ResampleFromInfostate(int player_id, std::function<double()> rng) const {
    const std::shared_ptr<const Game> game = batch_.GetGame();
    const int P = game->NumPlayers();
    if (game->MaxChanceOutcomes() == 0) return Clone();  // perfect information: the state itself
    std::vector<Action> kept;  // chance outcomes `player_id` knows: they stay unavailable to the redrawn deals
    {
      int chance_seen = 0;
      for (const auto& pa : history_) {
        if (pa.first != kChancePlayerId) continue;
        if (chance_seen == player_id || chance_seen >= P) kept.push_back(pa.second);
        ++chance_seen;
      }
    }
    std::unique_ptr<State> out = game->NewInitialState();
    int chance_seen = 0;
    for (const auto& pa : history_) {
      Action a = pa.second;
      if (pa.first == kChancePlayerId) {
        if (chance_seen != player_id && chance_seen < P) {
          std::vector<Action> pool;
          for (const auto& ap : out->ChanceOutcomes())
            if (std::find(kept.begin(), kept.end(), ap.first) == kept.end()) pool.push_back(ap.first);
          if (pool.empty()) SpielFatalError("ResampleFromInfostate: no consistent deal");
          a = pool[std::min(pool.size() - 1, static_cast<size_t>(rng() * pool.size()))];
        }
        ++chance_seen;
      }
      out->ApplyAction(a);
    }
    return out;
  }
  ActionsAndProbs ChanceOutcomes() const {
    std::vector<double> p = batch_.ChanceOutcomeProbs();
    ActionsAndProbs out;
    for (size_t o = 0; o < p.size(); ++o)
      if (p[o] > 0) out.emplace_back(static_cast<Action>(o), p[o]);
    return out;
  }
  void ApplyAction(Action a) {  // spiel.cc:441-451
    if (a == kInvalidAction) SpielFatalError("ApplyAction: kInvalidAction");
    const Player p = CurrentPlayer();
    batch_.ApplyActions({static_cast<int32_t>(a)});
    snap_ = Snapshot{};
    history_.push_back({p, a});
  }
  std::vector<float> ObservationTensor(Player player) const {  // spiel.cc:908-919 bounds check
    CheckPlayer(player);
    return batch_.ObservationTensor(player);
  }
  std::vector<float> InformationStateTensor(Player player) const {
    CheckPlayer(player);
    return batch_.InformationStateTensor(player);
  }
  // the caller-buffer forms (spiel.h:713-714, 693-694: absl::Span<float> values; any type with data() / size() here)
  template <class SpanLike, class = decltype(std::declval<SpanLike&>().data())>
  void ObservationTensor(Player player, SpanLike values) const {
    const std::vector<float> t = ObservationTensor(player);
    if (values.size() != t.size()) SpielFatalError("ObservationTensor: the buffer does not have ObservationTensorSize() entries");
    std::copy(t.begin(), t.end(), values.data());
  }
  template <class SpanLike, class = decltype(std::declval<SpanLike&>().data())>
  void InformationStateTensor(Player player, SpanLike values) const {
    const std::vector<float> t = InformationStateTensor(player);
    if (values.size() != t.size()) SpielFatalError("InformationStateTensor: the buffer does not have InformationStateTensorSize() entries");
    std::copy(t.begin(), t.end(), values.data());
  }
  void ObservationTensor(Player player, std::vector<float>* values) const { *values = ObservationTensor(player); }  // spiel.h:716-718
  void InformationStateTensor(Player player, std::vector<float>* values) const { *values = InformationStateTensor(player); }
  std::string InformationStateString(Player player) const {  // kuhn_poker.cc:285-288, leduc_poker.cc:517-520
    CheckPlayer(player);
    // the perfect-information games: the history (tic_tac_toe.cc:229-233, connect_four.cc:287-291, hex.cc:367-371)
    if (batch_.GetGame()->MaxChanceOutcomes() == 0 && batch_.GetGame()->InformationStateTensorSize() == 0) return HistoryString();
    char buf[512];
    if (osg_information_state_string(batch_.handle(), 0, player, buf, sizeof(buf)) < 0) SpielFatalError(osg_last_error());
    return buf;
  }
  std::string InformationStateString() const { return InformationStateString(CurrentPlayer()); }
  std::string ObservationString(Player player) const {  // the games' ObservationString / board ToString
    CheckPlayer(player);
    char buf[1024];
    if (osg_observation_string(batch_.handle(), 0, player, buf, sizeof(buf)) < 0) SpielFatalError(osg_last_error());
    return buf;
  }
  std::string ObservationString() const { return ObservationString(CurrentPlayer()); }
  std::string ToString() const {  // the games' ToString (e.g. leduc_poker.cc:463-496)
    char buf[1024];
    if (osg_state_string(batch_.handle(), 0, buf, sizeof(buf)) < 0) SpielFatalError(osg_last_error());
    return buf;
  }
  std::string ActionToString(Player player, Action action) const {  // spiel.h:386-392
    char buf[64];
    const int who = player == kChancePlayerId ? -1 : player;
    if (osg_action_string(batch_.handle(), 0, who, static_cast<int32_t>(action), buf, sizeof(buf)) < 0)
      SpielFatalError(osg_last_error());
    return buf;
  }
  std::string ActionToString(Action action) const { return ActionToString(CurrentPlayer(), action); }
  std::string HistoryString() const {  // spiel.h:700-702: "a, b, c"
    std::string out;
    for (size_t i = 0; i < history_.size(); ++i) out += (i ? ", " : "") + std::to_string(history_[i].second);
    return out;
  }
  std::unique_ptr<State> Clone() const { return std::unique_ptr<State>(new State(*this)); }
  std::unique_ptr<State> Child(Action a) const {  // spiel.h:737-744
    std::unique_ptr<State> c = Clone();
    c->ApplyAction(a);
    return c;
  }
  std::vector<Action> History() const {
    std::vector<Action> h;
    for (const auto& pa : history_) h.push_back(pa.second);
    return h;
  }
  struct PlayerAction {  // spiel.h:594-600
    Player player;
    Action action;
    bool operator==(const PlayerAction& o) const { return player == o.player && action == o.action; }
  };
  std::vector<PlayerAction> FullHistory() const {  // spiel.h:605
    std::vector<PlayerAction> h;
    for (const auto& pa : history_) h.push_back({pa.first, pa.second});
    return h;
  }
  // spiel.cc:432-439: the legal action that prints as action_str
  Action StringToAction(Player player, const std::string& action_str) const {
    for (Action a : LegalActions())
      if (action_str == ActionToString(player, a)) return a;
    SpielFatalError("Couldn't find an action matching " + action_str);
  }
  Action StringToAction(const std::string& action_str) const { return StringToAction(CurrentPlayer(), action_str); }
  // spiel.cc:947-964: true where every action before this state was a chance outcome and the state is not a chance node
  bool IsInitialNonChanceState() const {
    if (IsChanceNode()) return false;
    for (const auto& pa : history_)
      if (pa.first != kChancePlayerId) return false;
    return true;
  }
  double PlayerReward(Player p) const { return Rewards()[p]; }
  StateType GetType() const {  // spiel.h:808
    return IsTerminal() ? StateType::kTerminal : (IsChanceNode() ? StateType::kChance : StateType::kDecision);
  }
  // State::Serialize (spiel.cc:411-430): "starting_state=<ToJson of the starting position>" first when the state was
  // built from a struct (tic_tac_toe.cc:336, connect_four.cc:512), then the action history, one action per line.
  std::string Serialize() const {
    std::string out;
    if (!starting_state_str_.empty()) out += std::string(kSerializeStartingState) + starting_state_str_ + "\n";
    for (const auto& pa : history_) out += std::to_string(pa.second) + "\n";
    return history_.empty() ? out + "\n" : out;
  }
  // spiel.h:892, 1257-1262
  std::string StartingStateStr() const { return starting_state_str_; }
  inline std::unique_ptr<State> StartingState() const;
  // ---- the struct API (spiel.h:340-473, 728-734; tic_tac_toe.cc:178-213, connect_four.cc:224-275) ----
  std::unique_ptr<StateStruct> ToStruct() const {
    const std::string name = ShortName();
    const Player cur = CurrentPlayer();
    const std::string who = cur == 0 ? "x" : (cur == 1 ? "o" : DefaultPlayerString(cur));
    if (name == "tic_tac_toe") {
      auto rv = std::make_unique<tic_tac_toe::TicTacToeStateStruct>();
      rv->current_player = who;
      for (char ch : BoardCells()) rv->board.push_back(std::string(1, ch));
      return rv;
    }
    if (name == "connect_four") {
      auto rv = std::make_unique<connect_four::ConnectFourStateStruct>();
      const std::vector<int> shape = batch_.GetGame()->ObservationTensorShape();
      const int rows = shape[1], cols = shape[2];
      const std::string cells = BoardCells();
      rv->board.assign(rows, std::vector<std::string>(cols));
      for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) rv->board[r][c] = std::string(1, cells[r * cols + c]);
      rv->current_player = who;
      rv->is_terminal = IsTerminal();
      if (rv->is_terminal) {
        const std::vector<double> ret = Returns();
        rv->winner = ret[0] > 0 ? "x" : (ret[0] < 0 ? "o" : "draw");
      }
      return rv;
    }
    SpielFatalError("ToStruct is not implemented.");
  }
  std::string ToJson() const { return ToStruct()->ToJson(); }
  std::unique_ptr<ObservationStruct> ToObservationStruct(Player player) const {
    CheckPlayer(player);
    const std::string name = ShortName();
    if (name == "tic_tac_toe") return std::make_unique<tic_tac_toe::TicTacToeObservationStruct>(ToJson());
    if (name == "connect_four") return std::make_unique<connect_four::ConnectFourObservationStruct>(ToJson());
    SpielFatalError("ToObservationStruct not implemented!");
  }
  std::unique_ptr<ObservationStruct> ToObservationStruct() const { return ToObservationStruct(CurrentPlayer()); }
  std::unique_ptr<ActionStruct> ActionToStruct(Player /*player*/, Action action_id) const {
    const std::string name = ShortName();
    if (name == "tic_tac_toe") {
      auto a = std::make_unique<tic_tac_toe::TicTacToeActionStruct>();
      a->row = static_cast<int>(action_id) / tic_tac_toe::kNumCols;
      a->col = static_cast<int>(action_id) % tic_tac_toe::kNumCols;
      return a;
    }
    if (name == "connect_four") {
      auto a = std::make_unique<connect_four::ConnectFourActionStruct>();
      a->column = static_cast<int>(action_id);
      return a;
    }
    SpielFatalError("ActionToStruct not implemented.");
  }
  std::unique_ptr<ActionStruct> ActionToStruct(Action action_id) const { return ActionToStruct(CurrentPlayer(), action_id); }
  std::vector<Action> StructToActions(const ActionStruct& action_struct) const {
    const std::string name = ShortName();
    if (name == "tic_tac_toe") {
      const auto* a = SafeActionCast<tic_tac_toe::TicTacToeActionStruct>(action_struct);
      if (a->row < 0 || a->row >= tic_tac_toe::kNumRows || a->col < 0 || a->col >= tic_tac_toe::kNumCols)
        SpielFatalError("StructToActions: (row, col) is off the board");
      return {static_cast<Action>(a->row * tic_tac_toe::kNumCols + a->col)};
    }
    if (name == "connect_four") {
      const auto* a = SafeActionCast<connect_four::ConnectFourActionStruct>(action_struct);
      if (a->column < 0 || a->column >= batch_.GetGame()->ObservationTensorShape()[2])
        SpielFatalError("StructToActions: the column is off the board");
      return {static_cast<Action>(a->column)};
    }
    SpielFatalError("StructToActions not implemented.");
  }
  std::unique_ptr<ActionStruct> ActionsToStruct(Player player, const std::vector<Action>& actions) const {  // spiel.h:440-448
    if (actions.size() != 1) SpielFatalError("ActionsToStruct: one action per move in this game");
    return ActionToStruct(player, actions[0]);
  }
  std::unique_ptr<ActionStruct> ActionsToStruct(const std::vector<Action>& actions) const { return ActionsToStruct(CurrentPlayer(), actions); }
  Status ValidateActionStruct(const ActionStruct& action_struct) const {  // spiel.cc:491-504
    const std::vector<Action> legal = LegalActions();
    for (Action action : StructToActions(action_struct))
      if (std::find(legal.begin(), legal.end(), action) == legal.end())
        return ErrorStatus("Illegal action: " + action_struct.ToJson() + " (action " + std::to_string(action) + " = '" +
                           ActionToString(action) + "' is not legal)");
    return OkStatus();
  }
  Status ApplyActionStruct(const ActionStruct& action_struct) {  // spiel.cc:506-515
    Status status = ValidateActionStruct(action_struct);
    if (!status.ok()) return status;
    for (Action action : StructToActions(action_struct)) ApplyAction(action);
    return OkStatus();
  }
  // The position's cells ('.', 'x', 'o'): tic_tac_toe cell a = action a; connect_four cell r * cols + c, row 0 the bottom
  // row.  Read off the observation tensor (plane of x, plane of o), which the device packs from the same words.
  std::string BoardCells() const {
    const std::string name = ShortName();
    const std::vector<float> t = batch_.ObservationTensor(0);
    const std::vector<int> shape = batch_.GetGame()->ObservationTensorShape();
    const int cells = shape[1] * shape[2];
    std::string out(static_cast<size_t>(cells), '.');
    // tic_tac_toe planes: 0 empty, 1 o, 2 x (tic_tac_toe.cc:241-251); connect_four: 0 x, 1 o, 2 empty
    // (connect_four.cc:312-328; the egocentric form swaps the first two for player 1 — player 0 is asked here... and
    // for player 0 it holds o first, x second)
    int x_plane = name == "tic_tac_toe" ? 2 : 0, o_plane = 1;
    if (name == "connect_four") {
      const GameParameters params = GameParametersFromString(batch_.GetGame()->ToString());
      auto ego = params.find("egocentric_obs_tensor");   // (the values of a parsed game string are strings)
      const std::string text = ego == params.end() ? "" : ego->second.ToString();
      if (text == "True" || text == "true" || text == "1") { x_plane = 1; o_plane = 0; }
    }
    for (int i = 0; i < cells; ++i) {
      if (t[static_cast<size_t>(x_plane) * cells + i] != 0.0f) out[i] = 'x';
      else if (t[static_cast<size_t>(o_plane) * cells + i] != 0.0f) out[i] = 'o';
    }
    return out;
  }
  // This state := the position with these cells (osg_batch_set_cells); its history starts here, as the reference's
  // states constructed from a board do.
  void SetCells(const std::string& cells) {
    Check(osg_batch_set_cells(batch_.handle(), 0, cells.data(), static_cast<int>(cells.size())));
    snap_ = Snapshot{};
    history_.clear();
    starting_cells_ = cells;
    starting_state_str_.clear();   // (the struct constructors set it once their validation has passed)
  }
  // tic_tac_toe.cc:336, connect_four.cc:512: "Store the starting state for serialization"
  void RememberStartingState() { starting_state_str_ = ToJson(); }
  std::string ShortName() const {
    const std::string text = batch_.GetGame()->ToString();
    return text.substr(0, text.find('('));
  }
  int MoveNumber() const { return static_cast<int>(history_.size()); }
  int NumPlayers() const { return batch_.GetGame()->NumPlayers(); }
  std::shared_ptr<const Game> GetGame() const { return batch_.GetGame(); }
  const BatchedState& Batch() const { return batch_; }

 private:
  void CheckPlayer(Player player) const {
    if (player < 0 || player >= NumPlayers()) SpielFatalError("player id out of range");
  }
  struct Snapshot {
    bool have_status = false, have_mask = false;
    int8_t cur = 0;
    uint8_t term = 0;
    std::vector<double> returns;
    std::vector<uint32_t> bits;
  };
  const Snapshot& Snap() const {
    if (!snap_.have_status) {
      snap_.returns.assign(static_cast<size_t>(NumPlayers()), 0.0);
      Check(osg_status_query(batch_.handle(), &snap_.cur, &snap_.term, snap_.returns.data(), 1));
      snap_.have_status = true;
    }
    return snap_;
  }
  BatchedState batch_;
  mutable Snapshot snap_;
  std::vector<std::pair<Player, Action>> history_;
  std::string starting_cells_;       // the position SetCells installed ("" = the game's initial state): UndoAction's base
  std::string starting_state_str_;   // spiel.h:915
};

inline std::ostream& operator<<(std::ostream& stream, const State& state) { return stream << state.ToString(); }  // spiel.h:918
inline std::unique_ptr<State> Game::NewInitialState() const {
  return std::unique_ptr<State>(new State(shared_from_this()));
}
inline BatchedState Game::NewInitialStates(int64_t n) const { return BatchedState(shared_from_this(), n); }

// ---- the game-specific state / game classes user code names (tic_tac_toe.h:78-150, connect_four.h:115-200).  One
// State class serves every game here (a state is a record in a struct-of-arrays batch); these add constructors and
// accessors, no data, so a State* of the right game may be viewed as one (the reference's static_cast idiom). ----
namespace tic_tac_toe {
class TicTacToeState : public State {
 public:
  explicit TicTacToeState(std::shared_ptr<const Game> game) : State(std::move(game)) {}
  // tic_tac_toe.cc:273-336: the position of the struct, validated as the reference validates it
  TicTacToeState(std::shared_ptr<const Game> game, const TicTacToeStateStruct& state_struct) : State(std::move(game)) {
    if (static_cast<int>(state_struct.board.size()) != kNumCells)
      SpielFatalError("Invalid board size: expected " + std::to_string(kNumCells) + ", got " + std::to_string(state_struct.board.size()));
    std::string cells;
    int num_x = 0, num_o = 0;
    for (const std::string& cell : state_struct.board) {
      if (cell != "." && cell != "x" && cell != "o") SpielFatalError("Invalid cell value: '" + cell + "'. Expected '.', 'x', or 'o'.");
      cells.push_back(cell[0]);
      num_x += cell == "x";
      num_o += cell == "o";
    }
    if (num_x < num_o || num_x > num_o + 1)
      SpielFatalError("Invalid board state: invalid number of pieces, got x = " + std::to_string(num_x) + ", o = " + std::to_string(num_o));
    SetCells(cells);  // (both players with a line: refused there)
    if (IsTerminal()) {
      const double r0 = Returns()[0];
      if (r0 > 0 && num_x != num_o + 1)
        SpielFatalError("Invalid board state: x has a line, but number of pieces is inconsistent, got x = " + std::to_string(num_x) +
                        ", o = " + std::to_string(num_o));
      if (r0 < 0 && num_x != num_o)
        SpielFatalError("Invalid board state: o has a line, but number of pieces is inconsistent, got x = " + std::to_string(num_x) +
                        ", o = " + std::to_string(num_o));
    }
    const Player cur = CurrentPlayer();
    const std::string who = cur == 0 ? "x" : (cur == 1 ? "o" : DefaultPlayerString(cur));
    if (state_struct.current_player != who)
      SpielFatalError("Invalid current player: expected " + who + ", got " + state_struct.current_player);
    RememberStartingState();
  }
};
}  // namespace tic_tac_toe

namespace connect_four {
class ConnectFourGame : public Game {
 public:
  using Game::Game;
  using Game::NewInitialState;
  int rows() const { return Desc().obs_shape[1]; }
  int cols() const { return Desc().obs_shape[2]; }
  int x_in_row() const {
    const GameParameters params = GetParameters();
    auto it = params.find("x_in_row");
    return it == params.end() ? kDefaultXInRow : std::atoi(it->second.ToString().c_str());   // (parsed values are strings)
  }
  inline std::unique_ptr<State> NewInitialState(const ConnectFourStateStruct& state_struct, bool strict_validation = true) const;
};
class ConnectFourState : public State {
 public:
  explicit ConnectFourState(std::shared_ptr<const Game> game) : State(std::move(game)) {}
  // connect_four.cc:515-562: the board as ToString prints it (top row first; other characters are skipped)
  ConnectFourState(std::shared_ptr<const Game> game, const std::string& str) : State(std::move(game)) {
    const int rows = GetGame()->Desc().obs_shape[1], cols = GetGame()->Desc().obs_shape[2];
    std::string cells(static_cast<size_t>(rows) * cols, '.');
    int xs = 0, os = 0, r = rows - 1, c = 0;
    for (const char ch : str) {
      if (ch != '.' && ch != 'x' && ch != 'o') continue;
      if (r < 0) SpielFatalError("Problem parsing state (incorrect rows).");
      cells[static_cast<size_t>(r) * cols + c] = ch;
      xs += ch == 'x';
      os += ch == 'o';
      if (++c >= cols) { --r; c = 0; }
    }
    if (!(xs == os || xs == os + 1)) SpielFatalError("ConnectFourState: xs == os || xs == (os + 1)");
    if (r != -1) SpielFatalError("Problem parsing state (incorrect rows).");
    if (c != 0) SpielFatalError("Problem parsing state (column value should be 0)");
    SetCells(cells);
  }
  // connect_four.cc:352-513: the position of the struct; strict_validation = false admits unreachable piece counts,
  // everything else (no gaps, valid cells, terminal / winner / player consistency) is always checked
  ConnectFourState(std::shared_ptr<const Game> game, const ConnectFourStateStruct& state_struct, bool strict_validation = true)
      : State(std::move(game)) {
    const int rows = GetGame()->Desc().obs_shape[1], cols = GetGame()->Desc().obs_shape[2];
    if (static_cast<int>(state_struct.board.size()) != rows)
      SpielFatalError("Invalid board row count: expected " + std::to_string(rows) + ", got " + std::to_string(state_struct.board.size()));
    std::string cells;
    int num_x = 0, num_o = 0;
    for (int r = 0; r < rows; ++r) {
      if (static_cast<int>(state_struct.board[r].size()) != cols)
        SpielFatalError("Invalid board column count at row " + std::to_string(r) + ": expected " + std::to_string(cols) + ", got " +
                        std::to_string(state_struct.board[r].size()));
      for (const std::string& cell : state_struct.board[r]) {
        if (cell != "." && cell != "x" && cell != "o") SpielFatalError("Invalid cell value: '" + cell + "'. Expected '.', 'x', or 'o'.");
        cells.push_back(cell[0]);
        num_x += cell == "x";
        num_o += cell == "o";
      }
    }
    const std::string counts = "x=" + std::to_string(num_x) + ", o=" + std::to_string(num_o);
    if (strict_validation && (num_x < num_o || num_x > num_o + 1))
      SpielFatalError("Invalid board state: piece count imbalance. X (first player) must have equal or one more piece than O. Got " +
                      counts + ". Use strict_validation=false to allow unreachable positions.");
    SetCells(cells);  // (a gap in a column, both players with a line: refused there)
    const bool terminal = IsTerminal();
    std::string winner;
    if (terminal) {
      const double r0 = Returns()[0];
      winner = r0 > 0 ? "x" : (r0 < 0 ? "o" : "draw");
      if (strict_validation && winner == "x" && num_x != num_o + 1)
        SpielFatalError("Invalid board state: X has a winning line but piece counts are inconsistent. When X wins, X must have one "
                        "more piece than O. Got " + counts + ".");
      if (strict_validation && winner == "o" && num_x != num_o)
        SpielFatalError("Invalid board state: O has a winning line but piece counts are inconsistent. When O wins, X and O must have "
                        "equal pieces. Got " + counts + ".");
    }
    if (state_struct.is_terminal != terminal)
      SpielFatalError(std::string("Invalid is_terminal: struct says ") + (state_struct.is_terminal ? "terminal" : "non-terminal") +
                      " but board state is " + (terminal ? "terminal" : "non-terminal") + ".");
    if (state_struct.winner != winner)
      SpielFatalError("Invalid winner: struct says '" + state_struct.winner + "' but computed winner is '" + winner + "'.");
    if (terminal) {
      if (state_struct.current_player != DefaultPlayerString(kTerminalPlayerId))
        SpielFatalError("Invalid current_player for terminal state: expected '" + DefaultPlayerString(kTerminalPlayerId) + "', got '" +
                        state_struct.current_player + "'.");
      RememberStartingState();
      return;
    }
    if (state_struct.current_player != "x" && state_struct.current_player != "o")
      SpielFatalError("Invalid current_player: expected 'x' or 'o', got '" + state_struct.current_player + "'.");
    // The device derives the player to move from the stone count (the layout has no mover word): a struct that names
    // the other player — possible only with strict_validation = false — is not representable here.
    const std::string by_count = CurrentPlayer() == 0 ? "x" : "o";
    if (state_struct.current_player != by_count)
      SpielFatalError("Invalid current_player: with " + counts + " pieces, it should be " + by_count + "'s turn, but struct says '" +
                      state_struct.current_player + "'" + (strict_validation ? "." : " (a position whose mover differs from the stone "
                      "count's parity is not representable on the device)."));
    RememberStartingState();
  }
};
inline std::unique_ptr<State> ConnectFourGame::NewInitialState(const ConnectFourStateStruct& state_struct, bool strict_validation) const {
  return std::unique_ptr<State>(new ConnectFourState(shared_from_this(), state_struct, strict_validation));
}
}  // namespace connect_four

inline std::unique_ptr<State> Game::NewInitialState(const StateStruct& state_struct) const {
  if (const auto* t = dynamic_cast<const tic_tac_toe::TicTacToeStateStruct*>(&state_struct))
    return std::unique_ptr<State>(new tic_tac_toe::TicTacToeState(shared_from_this(), *t));
  if (const auto* c = dynamic_cast<const connect_four::ConnectFourStateStruct*>(&state_struct))
    return std::unique_ptr<State>(new connect_four::ConnectFourState(shared_from_this(), *c));
  SpielFatalError("NewInitialState from StateStruct is not implemented.");
}
inline std::unique_ptr<State> Game::NewInitialState(const std::string& str) const { return NewInitialState(Json::parse(str)); }
inline std::unique_ptr<State> Game::NewInitialState(const char* str) const { return NewInitialState(std::string(str)); }
inline std::unique_ptr<State> Game::NewInitialState(const Json& json) const {
  const std::string name = GetType().short_name;
  if (name == "tic_tac_toe") return NewInitialState(tic_tac_toe::TicTacToeStateStruct(json));
  if (name == "connect_four") return NewInitialState(connect_four::ConnectFourStateStruct(json));
  SpielFatalError("NewInitialState from JSON is not implemented.");
}
inline std::unique_ptr<State> State::StartingState() const {
  if (!starting_state_str_.empty()) return GetGame()->NewInitialState(Json::parse(starting_state_str_));
  return nullptr;
}
inline std::unique_ptr<State> Game::DeserializeState(const std::string& str) const {
  std::unique_ptr<State> state;
  size_t pos = 0;
  const size_t tag = std::strlen(kSerializeStartingState);
  if (str.compare(0, tag, kSerializeStartingState) == 0) {   // spiel.cc:550-558: the first line is the starting position
    size_t nl = str.find('\n');
    if (nl == std::string::npos) nl = str.size();
    state = NewInitialState(str.substr(tag, nl - tag));
    pos = nl + 1;
  } else {
    state = NewInitialState();
  }
  while (pos < str.size()) {
    size_t nl = str.find('\n', pos);
    if (nl == std::string::npos) nl = str.size();
    const std::string line = str.substr(pos, nl - pos);
    pos = nl + 1;
    if (line.empty()) continue;
    char* end = nullptr;
    const long long a = std::strtoll(line.c_str(), &end, 10);
    if (end == line.c_str() || *end != '\0') SpielFatalError("DeserializeState: not an action: " + line);
    state->ApplyAction(static_cast<Action>(a));  // illegal actions are fatal, as in the reference
  }
  return state;
}

// SerializeGameAndState / DeserializeGameAndState (spiel.cc:582-647): [Meta] / [Game] / [State].
constexpr const char* kSerializeStateSectionHeader = "[State]";
inline std::string SerializeGameAndState(const Game& game, const State& state) {
  return std::string("# Automatically generated by OpenSpiel SerializeGameAndState\n[Meta]\nVersion: 1\n\n[Game]\n") +
         game.Serialize() + "\n" + kSerializeStateSectionHeader + "\n" + state.Serialize() + "\n";
}
inline std::pair<std::shared_ptr<const Game>, std::unique_ptr<State>> DeserializeGameAndState(
    const std::string& serialized) {
  std::string sections[3];
  int current = -1;
  size_t pos = 0;
  while (pos <= serialized.size()) {
    size_t nl = serialized.find('\n', pos);
    if (nl == std::string::npos) nl = serialized.size();
    const std::string line = serialized.substr(pos, nl - pos);
    pos = nl + 1;
    if (line.empty() || line[0] == '#') continue;
    if (line == "[Meta]") { if (current != -1) SpielFatalError("malformed game-and-state text"); current = 0; }
    else if (line == "[Game]") { if (current != 0) SpielFatalError("malformed game-and-state text"); current = 1; }
    else if (line == kSerializeStateSectionHeader) { if (current != 1) SpielFatalError("malformed game-and-state text"); current = 2; }
    else if (current < 0) SpielFatalError("malformed game-and-state text");
    else sections[current] += line + "\n";
  }
  if (!sections[1].empty() && sections[1].back() == '\n') sections[1].pop_back();
  std::shared_ptr<const Game> game = LoadGame(sections[1]);
  std::unique_ptr<State> state = game->DeserializeState(sections[2]);
  return {game, std::move(state)};
}

// ---- observer.h: the observation API of the Python Observation class (python/observation.py) -------------
// Two observers exist on the device, the ones State::ObservationTensor and State::InformationStateTensor pack:
// the game's default observer and the perfect-recall, single-player-private one (kInfoStateObsType).
enum class PrivateInfoType { kNone, kSinglePlayer, kAllPlayers };  // observer.h:60-67
struct IIGObservationType {                                          // observer.h:75-104
  bool public_info = true;
  bool perfect_recall = false;
  PrivateInfoType private_info = PrivateInfoType::kSinglePlayer;
  bool operator==(const IIGObservationType& o) const {
    return public_info == o.public_info && perfect_recall == o.perfect_recall && private_info == o.private_info;
  }
};
constexpr IIGObservationType kDefaultObsType{true, false, PrivateInfoType::kSinglePlayer};    // observer.h:108-112
constexpr IIGObservationType kInfoStateObsType{true, true, PrivateInfoType::kSinglePlayer};   // observer.h:114-118

class SpanTensorInfo {  // observer.h:130-160
 public:
  SpanTensorInfo(std::string name, std::vector<int> shape) : name_(std::move(name)), shape_(std::move(shape)) {}
  const std::string& name() const { return name_; }
  const std::vector<int>& shape() const { return shape_; }
  std::vector<int> vector_shape() const { return shape_; }
  int size() const { int n = 1; for (int d : shape_) n *= d; return n; }
  std::string DebugString() const {
    std::string s = "SpanTensor(name='" + name_ + "', shape=(";
    for (size_t i = 0; i < shape_.size(); ++i) s += (i ? ", " : "") + std::to_string(shape_[i]);
    return s + "))";
  }

 private:
  std::string name_;
  std::vector<int> shape_;
};
class SpanTensor {  // observer.h:162-187: a named view into the Observation's buffer
 public:
  SpanTensor(SpanTensorInfo info, float* data) : info_(std::move(info)), data_(data) {}
  const SpanTensorInfo& info() const { return info_; }
  float* data() const { return data_; }
  std::string DebugString() const { return info_.DebugString(); }

 private:
  SpanTensorInfo info_;
  float* data_;
};

class Observer {  // observer.h:280-305; made by Game::MakeObserver
 public:
  // kDefault / kInfoState: what State::ObservationTensor / InformationStateTensor pack on the device.  kGeneral: any other
  // IIGObservationType of the two poker games (kuhn_poker.cc:65-165, leduc_poker.cc:92-242): the same pieces, chosen and
  // arranged as the type asks — composed on the host from the two device tensors (private_info = kAllPlayers: every
  // player's card piece).  kNoPrivate: a perfect-information game asked for private information only (observer.cc:110-121):
  // an empty tensor and an empty string.  kInfoStateString: a board game's perfect-recall observer (observer.cc:158-159):
  // the information-state STRING, no tensor.
  enum class Kind { kDefault, kInfoState, kGeneral, kNoPrivate, kInfoStateString };
  Observer(bool info_state, std::vector<SpanTensorInfo> pieces)
      : kind_(info_state ? Kind::kInfoState : Kind::kDefault), type_(info_state ? kInfoStateObsType : kDefaultObsType),
        pieces_(std::move(pieces)) {}
  Observer(Kind kind, IIGObservationType type, std::vector<SpanTensorInfo> pieces)
      : kind_(kind), type_(type), pieces_(std::move(pieces)) {}
  bool HasString() const { return true; }
  bool HasTensor() const { return kind_ != Kind::kInfoStateString; }
  bool info_state() const { return kind_ == Kind::kInfoState; }
  Kind kind() const { return kind_; }
  const IIGObservationType& type() const { return type_; }
  const std::vector<SpanTensorInfo>& pieces() const { return pieces_; }

 private:
  Kind kind_;
  IIGObservationType type_;
  std::vector<SpanTensorInfo> pieces_;
};

// The tensor pieces in the order the reference's observers write them (kuhn_poker.cc:72-107,
// leduc_poker.cc:103-192; DefaultObserver for the board games), or empty when the type is not offered.
inline std::vector<SpanTensorInfo> ObserverPieces(const Game& game, bool info_state) {
  const std::string text = game.ToString();
  const std::string name = text.substr(0, text.find('('));
  const int P = game.NumPlayers();
  if (name == "kuhn_poker") {
    std::vector<SpanTensorInfo> v{{"player", {P}}, {"private_card", {P + 1}}};
    if (info_state) v.push_back({"betting", {2 * P - 1, 2}}); else v.push_back({"pot_contribution", {P}});
    return v;
  }
  if (name == "leduc_poker") {
    const int cards = (game.ObservationTensorSize() - 2 * P) / 2;
    std::vector<SpanTensorInfo> v{{"player", {P}}, {"private_card", {cards}}, {"community_card", {cards}}};
    if (info_state) v.push_back({"betting", {2, 3 * P - 2, 2}}); else v.push_back({"pot_contribution", {P}});
    return v;
  }
  if (info_state) return {};
  return {{"observation", game.ObservationTensorShape()}};
}
// The pieces of any IIGObservationType of the two poker games (kuhn_poker.cc:72-107: nothing private unless
// kSinglePlayer; leduc_poker.cc:166-186: the observing player always, one card or every player's cards, then the public part).
inline std::vector<SpanTensorInfo> GeneralObserverPieces(const Game& game, const IIGObservationType& t) {
  const std::string text = game.ToString();
  const std::string name = text.substr(0, text.find('('));
  const int P = game.NumPlayers();
  std::vector<SpanTensorInfo> v;
  if (name == "kuhn_poker") {
    if (t.private_info == PrivateInfoType::kSinglePlayer) { v.push_back({"player", {P}}); v.push_back({"private_card", {P + 1}}); }
    if (t.public_info) {
      if (t.perfect_recall) v.push_back({"betting", {2 * P - 1, 2}}); else v.push_back({"pot_contribution", {P}});
    }
  } else if (name == "leduc_poker") {
    const int cards = (game.ObservationTensorSize() - 2 * P) / 2;
    v.push_back({"player", {P}});
    if (t.private_info == PrivateInfoType::kSinglePlayer) v.push_back({"private_card", {cards}});
    else if (t.private_info == PrivateInfoType::kAllPlayers) v.push_back({"private_cards", {P, cards}});
    if (t.public_info) {
      v.push_back({"community_card", {cards}});
      if (t.perfect_recall) v.push_back({"betting", {2, 3 * P - 2, 2}}); else v.push_back({"pot_contribution", {P}});
    }
  }
  return v;
}

inline std::shared_ptr<Observer> MakeObserver(const Game& game, const IIGObservationType* iig_obs_type = nullptr) {
  // spiel.h:1040-1054, observer.cc:137-174: null = the game's default observer
  const bool poker = game.MaxChanceOutcomes() > 0;
  if (iig_obs_type && !poker) {  // perfect-information games (observer.cc:150-161)
    if (!iig_obs_type->public_info) return std::make_shared<Observer>(Observer::Kind::kNoPrivate, *iig_obs_type, std::vector<SpanTensorInfo>{});
    if (iig_obs_type->perfect_recall)
      return std::make_shared<Observer>(Observer::Kind::kInfoStateString, *iig_obs_type, std::vector<SpanTensorInfo>{});
    return std::make_shared<Observer>(false, ObserverPieces(game, false));
  }
  bool info_state = false;
  if (iig_obs_type) {
    if (*iig_obs_type == kInfoStateObsType) info_state = true;
    else if (!(*iig_obs_type == kDefaultObsType))   // the poker games' observers take any type (MakeObserver of both games)
      return std::make_shared<Observer>(Observer::Kind::kGeneral, *iig_obs_type, GeneralObserverPieces(game, *iig_obs_type));
  }
  std::vector<SpanTensorInfo> pieces = ObserverPieces(game, info_state);
  if (pieces.empty()) return nullptr;
  return std::make_shared<Observer>(info_state, std::move(pieces));
}

inline std::shared_ptr<Observer> Game::MakeObserver(std::optional<IIGObservationType> iig_obs_type,
                                                    const GameParameters& params) const {
  auto name = params.find("name");  // spiel.cc:866-879: a named observer from the registry, else the built-in one
  if (name != params.end() && name->second.string_value() != "single_tensor")
    SpielFatalError("No observer '" + name->second.string_value() + "' found for game '" + GetType().short_name + "'");
  return open_spiel::hip::MakeObserver(*this, iig_obs_type ? &*iig_obs_type : nullptr);
}

class Observation {  // observer.h:309-371: owns the flat buffer the observer writes into
 public:
  Observation(const Game& game, std::shared_ptr<Observer> observer) : observer_(std::move(observer)) {
    if (!observer_) SpielFatalError("Observation: null observer");
    int total = 0;
    for (const SpanTensorInfo& p : observer_->pieces()) total += p.size();
    if (observer_->kind() == Observer::Kind::kDefault || observer_->kind() == Observer::Kind::kInfoState) {
      const int expect = observer_->info_state() ? game.InformationStateTensorSize() : game.ObservationTensorSize();
      if (total != expect) SpielFatalError("Observation: piece layout does not match the game's tensor");
    }
    buffer_.assign(static_cast<size_t>(total), 0.0f);
  }
  std::vector<float>& Tensor() { return buffer_; }
  std::vector<SpanTensorInfo> tensors_info() const { return observer_->pieces(); }
  std::vector<SpanTensor> tensors() {
    std::vector<SpanTensor> out;
    int offset = 0;
    for (const SpanTensorInfo& p : observer_->pieces()) {
      out.emplace_back(p, buffer_.data() + offset);
      offset += p.size();
    }
    return out;
  }
  void SetFrom(const State& state, int player) {
    switch (observer_->kind()) {
      case Observer::Kind::kDefault:
      case Observer::Kind::kInfoState: {
        const std::vector<float> v = observer_->info_state() ? state.InformationStateTensor(player) : state.ObservationTensor(player);
        std::copy(v.begin(), v.end(), buffer_.begin());
        return;
      }
      case Observer::Kind::kNoPrivate:
      case Observer::Kind::kInfoStateString:
        return;
      case Observer::Kind::kGeneral:
        SetGeneral(state, player);
        return;
    }
  }
  std::string StringFrom(const State& state, int player) const {
    switch (observer_->kind()) {
      case Observer::Kind::kDefault: return state.ObservationString(player);
      case Observer::Kind::kInfoState:
      case Observer::Kind::kInfoStateString: return state.InformationStateString(player);
      case Observer::Kind::kNoPrivate: return "";
      case Observer::Kind::kGeneral: break;
    }
    return GeneralString(state, player);
  }
  bool HasString() const { return observer_->HasString(); }
  bool HasTensor() const { return observer_->HasTensor(); }

  // observer.cc:246-321: one header byte (0 = raw floats, 1 = one bit per element when every element is 0 or 1)
  std::string Compress() const {
    bool binary = true;
    for (float x : buffer_) binary &= (x == 0.0f || x == 1.0f);
    if (!binary) {
      std::string out(1 + sizeof(float) * buffer_.size(), '\0');
      if (!buffer_.empty()) std::memcpy(&out[1], buffer_.data(), sizeof(float) * buffer_.size());
      return out;
    }
    std::string out(1 + (buffer_.size() + 7) / 8, '\0');
    out[0] = 1;
    for (size_t i = 0; i < buffer_.size(); ++i)
      if (buffer_[i] != 0.0f) out[1 + i / 8] = static_cast<char>(out[1 + i / 8] + (1 << (i % 8)));
    return out;
  }
  void Decompress(const std::string& compressed) {
    if (compressed.empty()) SpielFatalError("Decompress: empty string");
    if (compressed[0] == 1) {
      if (compressed.size() != 1 + (buffer_.size() + 7) / 8) SpielFatalError("Decompress: size does not match the observation");
      for (size_t i = 0; i < buffer_.size(); ++i) buffer_[i] = (compressed[1 + i / 8] >> (i % 8)) & 1 ? 1.0f : 0.0f;
    } else if (compressed[0] == 0) {
      if (compressed.size() != 1 + sizeof(float) * buffer_.size()) SpielFatalError("Decompress: size does not match the observation");
      if (!buffer_.empty()) std::memcpy(buffer_.data(), &compressed[1], sizeof(float) * buffer_.size());
    } else {
      SpielFatalError("Unrecognized compression scheme in '" + compressed + "'");
    }
  }

 private:
  // a named piece of one of the two device tensors of `player`
  static std::vector<float> Piece(const State& state, int player, bool info_state, const std::string& name) {
    const std::vector<float> t = info_state ? state.InformationStateTensor(player) : state.ObservationTensor(player);
    int offset = 0;
    for (const SpanTensorInfo& p : ObserverPieces(*state.GetGame(), info_state)) {
      if (p.name() == name) return std::vector<float>(t.begin() + offset, t.begin() + offset + p.size());
      offset += p.size();
    }
    SpielFatalError("Observation: no piece '" + name + "'");
  }
  void SetGeneral(const State& state, int player) {
    const int P = state.NumPlayers();
    int offset = 0;
    for (const SpanTensorInfo& p : observer_->pieces()) {
      std::vector<float> v;
      if (p.name() == "private_cards") {
        for (int q = 0; q < P; ++q) {
          const std::vector<float> one = Piece(state, q, false, "private_card");
          v.insert(v.end(), one.begin(), one.end());
        }
      } else {
        v = Piece(state, player, p.name() == "betting", p.name());
      }
      std::copy(v.begin(), v.end(), buffer_.begin() + offset);
      offset += p.size();
    }
  }
  static std::vector<std::string> Chunks(const std::string& text) {  // "[a][b]" -> {"[a]", "[b]"}
    std::vector<std::string> out;
    size_t pos = 0;
    while (pos < text.size()) {
      const size_t end = text.find(']', pos);
      if (end == std::string::npos) break;
      out.push_back(text.substr(pos, end + 1 - pos));
      pos = end + 1;
    }
    return out;
  }
  static std::string ChunkWith(const std::vector<std::string>& chunks, const std::string& prefix) {
    for (const std::string& c : chunks)
      if (c.compare(0, prefix.size(), prefix) == 0) return c;
    return "";
  }
  std::string GeneralString(const State& state, int player) const {
    const IIGObservationType& t = observer_->type();
    const std::string text = state.GetGame()->ToString();
    const std::string name = text.substr(0, text.find('('));
    const int P = state.NumPlayers();
    std::string result;
    if (name == "kuhn_poker") {  // kuhn_poker.cc:109-165
      const std::vector<Action> h = state.History();
      const int n = static_cast<int>(h.size());
      if (t.private_info == PrivateInfoType::kSinglePlayer) {
        if (t.perfect_recall || t.public_info) {
          if (n > player) result += std::to_string(h[player]);
        } else if (n == 1 + player) {
          result += "Received card " + std::to_string(h[player]);
        }
      }
      if (t.public_info) {
        if (t.perfect_recall) {
          for (int i = P; i < n; ++i) result.push_back(h[i] ? 'b' : 'p');
        } else if (t.private_info == PrivateInfoType::kNone) {
          if (n == 0) result += "start game";
          else if (n > P) result += h.back() ? "Bet" : "Pass";
        } else if (n > player) {
          for (float ante : Piece(state, player, false, "pot_contribution")) result += std::to_string(static_cast<int>(ante));
        }
      }
      if (t.public_info && t.private_info == PrivateInfoType::kNone && n > 0 && n <= P)
        result += "Deal to player " + std::to_string(n - 1);
      return result;
    }
    // leduc_poker.cc:192-238: the chunks of the two device-formatted strings, chosen as the type asks
    const std::vector<std::string> obs = Chunks(state.ObservationString(player));
    if (t.private_info == PrivateInfoType::kSinglePlayer) {
      result += ChunkWith(obs, "[Observer: ") + ChunkWith(obs, "[Private: ");
    } else if (t.private_info == PrivateInfoType::kAllPlayers) {
      result += "[Privates: ";
      for (int q = 0; q < P; ++q) {
        const std::string c = ChunkWith(Chunks(state.ObservationString(q)), "[Private: ");
        result += c.substr(10, c.size() - 11);
      }
      result += "]";
    }
    if (t.public_info) {
      result += ChunkWith(obs, "[Round ") + ChunkWith(obs, "[Player: ") + ChunkWith(obs, "[Pot: ") + ChunkWith(obs, "[Money: ") +
                ChunkWith(obs, "[Public: ");
      if (t.perfect_recall) {
        const std::vector<std::string> info = Chunks(state.InformationStateString(player));
        result += ChunkWith(info, "[Round1: ") + ChunkWith(info, "[Round2: ");
      } else {
        result += ChunkWith(obs, "[Ante: ");
      }
    }
    return result;
  }

  std::shared_ptr<Observer> observer_;
  std::vector<float> buffer_;
};

namespace algorithms {

class Evaluator {  // mcts.h:83-92
 public:
  virtual ~Evaluator() = default;
  virtual std::vector<double> Evaluate(const State& state) = 0;
  virtual ActionsAndProbs Prior(const State& state) = 0;
};

class RandomRolloutEvaluator : public Evaluator {  // mcts.h:97-111
 public:
  RandomRolloutEvaluator(int n_rollouts, int seed) : n_rollouts_(n_rollouts), seed_(seed) {}
  std::vector<double> Evaluate(const State& state) override {  // mcts.cc:43-72
    std::vector<double> sum = EvaluateBatch(state.Batch());
    return sum;
  }
  // Mean returns [n, P] of n_rollouts uniform-random playouts from every state of the batch.
  std::vector<double> EvaluateBatch(const BatchedState& states) {
    std::vector<double> sum(static_cast<size_t>(states.size()) * states.GetGame()->NumPlayers());
    Check(osg_rollout(states.handle(), static_cast<uint64_t>(seed_), calls_, n_rollouts_, sum.data(), nullptr, 1));
    calls_ += states.size();  // fresh streams for the next call, like an advancing mt19937
    for (double& v : sum) v /= n_rollouts_;
    return sum;
  }
  ActionsAndProbs Prior(const State& state) override {  // mcts.cc:74-87
    if (state.IsChanceNode()) return state.ChanceOutcomes();
    std::vector<Action> legal = state.LegalActions();
    ActionsAndProbs prior;
    for (Action a : legal) prior.emplace_back(a, 1.0 / legal.size());
    return prior;
  }
  int n_rollouts() const { return n_rollouts_; }
  int seed() const { return seed_; }

 private:
  int n_rollouts_;
  int seed_;
  int64_t calls_ = 0;
};

struct SearchNode {  // mcts.h:114-146
  Action action = kInvalidAction;
  double prior = 0;
  Player player = 0;
  int explore_count = 0;
  double total_reward = 0;
  std::vector<double> outcome;
  std::vector<SearchNode> children;
  bool CompareFinal(const SearchNode& b) const {  // mcts.cc:114-125
    // player is kChancePlayerId (-1) for the children of a chance node: the reference reads the outcome only
    // for 0 <= player < outcome.size() (mcts.cc:115-118)
    double mine = (outcome.empty() || player < 0 || player >= static_cast<Player>(outcome.size())) ? 0 : outcome[player];
    double theirs = (b.outcome.empty() || b.player < 0 || b.player >= static_cast<Player>(b.outcome.size())) ? 0 : b.outcome[b.player];
    if (mine != theirs) return mine < theirs;
    if (explore_count != b.explore_count) return explore_count < b.explore_count;
    return total_reward < b.total_reward;
  }
  const SearchNode& BestChild() const {  // mcts.cc:127-143
    const SearchNode* best = &children.front();
    for (const SearchNode& c : children)
      if (best->CompareFinal(c)) best = &c;
    return *best;
  }
  std::string ToString(const State& state) const {  // mcts.cc:165-179
    char buf[256];
    const std::string act = action != kInvalidAction ? state.ActionToString(player, action) : "none";
    std::string out_s = "none";
    if (!outcome.empty()) {
      char o[32];
      std::snprintf(o, sizeof(o), "%4.1f", outcome[player == kChancePlayerId ? 0 : player]);
      out_s = o;
    }
    std::snprintf(buf, sizeof(buf), "%6s: player: %d, prior: %5.3f, value: %6.3f, sims: %5d, outcome: %s, %3d children",
                  act.c_str(), player, prior, explore_count ? total_reward / explore_count : 0., explore_count,
                  out_s.c_str(), static_cast<int>(children.size()));
    return buf;
  }
  std::string ChildrenStr(const State& state) const {  // mcts.cc:146-163: best first
    std::vector<const SearchNode*> refs;
    for (const SearchNode& c : children) refs.push_back(&c);
    std::stable_sort(refs.begin(), refs.end(), [](const SearchNode* a, const SearchNode* b) { return b->CompareFinal(*a); });
    std::string out;
    for (const SearchNode* c : refs) out += c->ToString(state) + "\n";
    return out;
  }
};

enum class ChildSelectionPolicy { UCT, PUCT };  // mcts.h:148

}  // namespace algorithms

class Bot {  // spiel_bots.h:73-185
 public:
  virtual ~Bot() = default;
  virtual Action Step(const State& state) = 0;
  virtual std::pair<Action, std::string> StepVerbose(const State& state) { return {Step(state), ""}; }
  virtual void InformAction(const State&, Player, Action) {}
  virtual void InformActions(const State&, const std::vector<Action>&) {}
  virtual void Restart() {}
  virtual void RestartAt(const State&) { SpielFatalError("RestartAt(state) not implemented."); }
  virtual bool ProvidesForceAction() { return false; }
  virtual void ForceAction(const State&, Action) {
    SpielFatalError(ProvidesForceAction() ? "ForceAction not implemented but should because the bot is registered as overridable."
                                          : "ForceAction not implemented because the bot is not overridable");
  }
  virtual bool ProvidesPolicy() { return false; }
  virtual ActionsAndProbs GetPolicy(const State&) {
    SpielFatalError(ProvidesPolicy() ? "GetPolicy not implemented but should because the bot is registered as exposing its policy."
                                     : "GetPolicy not implemented because the bot is not exposing any policy.");
  }
  virtual std::pair<ActionsAndProbs, Action> StepWithPolicy(const State&) {
    SpielFatalError(ProvidesPolicy() ? "StepWithPolicy not implemented but should because the bot is registered as exposing its policy."
                                     : "StepWithPolicy not implemented because the bot is not exposing any policy.");
  }
  virtual bool IsClonable() const { return false; }
  virtual std::unique_ptr<Bot> Clone() { SpielFatalError("Clone method not implemented."); }
};

// spiel_utils.h SampleAction(outcomes, rng): the outcome whose cumulative-probability interval holds a uniform draw
inline std::pair<Action, double> SampleAction(const ActionsAndProbs& outcomes, std::mt19937& rng) {
  const double z = std::uniform_real_distribution<double>(0.0, 1.0)(rng);
  double acc = 0;
  for (const auto& ap : outcomes) {
    if (z >= acc && z < acc + ap.second) return ap;
    acc += ap.second;
  }
  return outcomes.back();
}

// algorithms/evaluate_bots.cc:27-62: play one episode from `state` with one bot per player, return the returns.
inline std::vector<double> EvaluateBots(State* state, const std::vector<Bot*>& bots, int seed) {
  std::mt19937 rng(seed);
  if (state->History().empty()) {
    for (Bot* bot : bots) bot->Restart();
  } else {
    for (Bot* bot : bots) bot->RestartAt(*state);
  }
  while (!state->IsTerminal()) {
    if (state->IsChanceNode()) {
      const Action action = SampleAction(state->ChanceOutcomes(), rng).first;
      for (Bot* bot : bots) bot->InformAction(*state, kChancePlayerId, action);
      state->ApplyAction(action);
    } else {
      const Player current = state->CurrentPlayer();
      const Action action = bots[current]->Step(*state);
      for (size_t p = 0; p < bots.size(); ++p)
        if (static_cast<Player>(p) != current) bots[p]->InformAction(*state, current, action);
      state->ApplyAction(action);
    }
  }
  return state->Returns();
}

namespace algorithms {

inline std::vector<double> dirichlet_noise(int count, double alpha, std::mt19937* rng) {  // mcts.cc:188-203
  std::vector<double> noise;
  std::gamma_distribution<double> gamma(alpha, 1.0);
  for (int i = 0; i < count; ++i) noise.push_back(gamma(*rng));
  double sum = 0;
  for (double v : noise) sum += v;
  for (double& v : noise) v /= sum;
  return noise;
}

// mcts.h:149-220.  The search runs on the device: StepBatch searches a whole batch of roots with the fused
// kernels (RandomRolloutEvaluator only); MCTSearch / Step run one root through the persistent-tree entry
// points (osg_mcts_tree_*), which serve ANY Evaluator — requests for Prior(state) and Evaluate(state) come back
// to the host, RandomRolloutEvaluator is answered on the device — and return the whole SearchNode tree.
class MCTSBot : public Bot {
 public:
  MCTSBot(const Game& game, std::shared_ptr<Evaluator> evaluator, double uct_c, int max_simulations,
          int64_t max_memory_mb, bool solve, int seed, bool verbose,
          ChildSelectionPolicy child_selection_policy = ChildSelectionPolicy::UCT, double dirichlet_alpha = 0,
          double dirichlet_epsilon = 0, bool dont_return_chance_node = false, double max_wall_clock_time = -1)
      : evaluator_(std::move(evaluator)), uct_c_(uct_c), max_simulations_(max_simulations),
        max_memory_mb_(max_memory_mb), solve_(solve), seed_(seed), verbose_(verbose),
        num_actions_(game.NumDistinctActions()), num_players_(game.NumPlayers()), policy_(child_selection_policy),
        dirichlet_alpha_(dirichlet_alpha), dirichlet_epsilon_(dirichlet_epsilon),
        dont_return_chance_node_(dont_return_chance_node), max_wall_clock_time_(max_wall_clock_time), rng_(seed) {
    rollout_ = dynamic_cast<RandomRolloutEvaluator*>(evaluator_.get());
    if (!evaluator_) SpielFatalError("MCTSBot needs an evaluator");
  }
  void Restart() override {}
  void RestartAt(const State&) override {}
  // true when the evaluator is the C++ RandomRolloutEvaluator: a search then never calls back into the host
  // language (the pybind module releases the GIL around Step for such bots, as bots.cc:147-148 does)
  bool EvaluatorIsNative() const { return rollout_ != nullptr; }
  // One search per state of the batch with the fused kernels; returns BestChild().action per root (-1 for
  // terminal roots).  RandomRolloutEvaluator, no root noise.
  std::vector<Action> StepBatch(const BatchedState& roots) {
    if (!rollout_) SpielFatalError("StepBatch: the fused batch kernels implement RandomRolloutEvaluator");
    std::vector<int32_t> best(roots.size());
    osg_mcts_cfg cfg = Config();
    Check(osg_mcts_search(roots.handle(), &cfg, best.data(), nullptr, nullptr, nullptr, nullptr, 1));
    searches_ += roots.size();
    return std::vector<Action>(best.begin(), best.end());
  }
  Action Step(const State& state) override {  // mcts.cc:233-266
    std::unique_ptr<SearchNode> root = MCTSearch(state);
    if (max_simulations_ <= 1 || root->children.empty()) {  // sample from the prior (mcts.cc:237-239)
      ActionsAndProbs prior = evaluator_->Prior(state);
      const double z = std::uniform_real_distribution<double>(0.0, 1.0)(rng_);
      double acc = 0;
      for (const auto& ap : prior) {
        if (z >= acc && z < acc + ap.second) return ap.first;
        acc += ap.second;
      }
      return prior.back().first;
    }
    const SearchNode& best = root->BestChild();
    if (verbose_) {
      std::fprintf(stderr, "Finished %d sims, tree size: %lld nodes.\nRoot:\n%s\nChildren:\n%s\n", root->explore_count,
                   static_cast<long long>(last_nodes_), root->ToString(state).c_str(), root->ChildrenStr(state).c_str());
    }
    return best.action;
  }
  std::pair<ActionsAndProbs, Action> StepWithPolicy(const State& state) override {  // mcts.cc:268-271
    const Action action = Step(state);
    return {{{action, 1.0}}, action};
  }
  bool ProvidesPolicy() override { return true; }
  ActionsAndProbs GetPolicy(const State& state) override { return StepWithPolicy(state).first; }  // spiel_bots.h:141

  std::unique_ptr<SearchNode> MCTSearch(const State& state) {  // mcts.cc:353-467
    if (max_simulations_ < 1 && max_wall_clock_time_ <= 0) {
      // the reference's loop body never runs (mcts.cc:362-366): the root alone, one visit, no children
      last_nodes_ = 1;
      auto root = std::make_unique<SearchNode>();
      root->player = state.CurrentPlayer();
      root->prior = 1.0;
      return root;
    }
    osg_mcts_cfg cfg = Config();
    const bool host_priors = rollout_ == nullptr || dirichlet_alpha_ > 0;
    // RandomRolloutEvaluator: the leaves are evaluated inside the launch (flag 4) — with its uniform prior a whole
    // search is ONE kernel launch; any other evaluator is asked through the request / answer rounds below
    const int flags = (host_priors ? 1 : 0) | (dont_return_chance_node_ ? 2 : 0) | (rollout_ ? 4 : 0);
    osg_mcts_tree* tree = nullptr;
    Check(osg_mcts_tree_create(state.Batch().handle(), &cfg, flags, &tree));
    struct Guard { osg_mcts_tree* t; ~Guard() { osg_mcts_tree_destroy(t); } } guard{tree};
    BatchedState leaf(state.GetGame(), 1);
    std::vector<double> prior(num_actions_), value(num_players_);
    const auto start = std::chrono::steady_clock::now();
    // max_wall_clock_time > 0 replaces the simulation bound (mcts.cc:362-366): the search runs in slices of
    // simulations and the clock is read between them
    const int slice = max_wall_clock_time_ > 0 ? 64 : (1 << 30);
    bool have_prior = false, have_value = false, device_value = false;
    for (;;) {
      int64_t counts[4];
      Check(osg_mcts_tree_advance_host(tree, leaf.handle(), have_prior ? prior.data() : nullptr,
                                       have_value ? value.data() : nullptr, device_value ? 1 : 0, nullptr, slice, counts));
      have_prior = have_value = device_value = false;
      if (counts[0] == 1) break;  // finished: the simulation bound, a proven root or a single root child
      if (max_wall_clock_time_ > 0 &&
          std::chrono::duration<double>(std::chrono::steady_clock::now() - start).count() >= max_wall_clock_time_)
        break;
      if (counts[3]) continue;    // paused between two slices: nothing wanted
      if (counts[2] && rollout_) {  // (only without flag 4) RandomRolloutEvaluator::Evaluate on the device
        Check(osg_mcts_tree_rollout_values(tree, leaf.handle(), nullptr));
        device_value = true;
        continue;
      }
      // the state the request is about, with its history: the root state advanced along the parked path
      int32_t path[256];
      const int len = osg_mcts_tree_leaf_path(tree, 0, path, 256);
      if (len < 0) SpielFatalError(osg_last_error());
      std::unique_ptr<State> at = state.Clone();
      for (int k = 0; k < len; ++k) at->ApplyAction(path[k]);
      if (counts[1]) {
        ActionsAndProbs ap = evaluator_->Prior(*at);
        if (len == 0 && dirichlet_alpha_ > 0) {  // root noise (mcts.cc:284-292)
          std::vector<double> noise = dirichlet_noise(static_cast<int>(ap.size()), dirichlet_alpha_, &rng_);
          for (size_t k = 0; k < ap.size(); ++k)
            ap[k].second = (1 - dirichlet_epsilon_) * ap[k].second + dirichlet_epsilon_ * noise[k];
        }
        std::fill(prior.begin(), prior.end(), 0.0);
        for (const auto& x : ap)
          if (x.first >= 0 && x.first < num_actions_) prior[x.first] = x.second;
        have_prior = true;
      } else {
        value = evaluator_->Evaluate(*at);
        if (static_cast<int>(value.size()) != num_players_) SpielFatalError("Evaluate() must return one value per player");
        have_value = true;
      }
    }
    ++searches_;
    return Download(tree, state);
  }

 private:
  std::unique_ptr<SearchNode> Download(osg_mcts_tree* tree, const State& state) {
    const int64_t used = osg_mcts_tree_nodes(tree, 0);
    if (used < 1) SpielFatalError(osg_last_error());
    last_nodes_ = used;
    std::vector<uint32_t> meta(used), first(used), count(used);
    std::vector<double> total(used), prior(used);
    Check(osg_mcts_tree_download(tree, 0, used, meta.data(), first.data(), count.data(), total.data(), prior.data()));
    const bool board = num_players_ == 2 && state.GetGame()->MaxChanceOutcomes() == 0;
    return SearchTreeFromArrays(meta, first, count, total, prior, state.CurrentPlayer(), num_players_, board);
  }

 public:
  // The device's flat tree (include/osg_abi.h osg_mcts_tree_download) as SearchNodes.  Pure host code: tested
  // without a device (tests/native/host_mirror_cpu_test.cpp).
  static std::unique_ptr<SearchNode> SearchTreeFromArrays(const std::vector<uint32_t>& meta, const std::vector<uint32_t>& first,
                                                          const std::vector<uint32_t>& count, const std::vector<double>& total,
                                                          const std::vector<double>& prior, Player root_player,
                                                          int num_players, bool board) {
    auto root = std::unique_ptr<SearchNode>(new SearchNode);
    std::vector<std::pair<SearchNode*, uint32_t>> todo{{root.get(), 0u}};
    while (!todo.empty()) {
      auto [node, i] = todo.back();
      todo.pop_back();
      const uint32_t m = meta[i];
      // (games with more than 255 actions keep the ninth bit of the action in bit 24 and of the child count in bit 25:
      // csrc/osg_mcts_internal.h; both are zero for every other game)
      node->action = i == 0 ? kInvalidAction : static_cast<Action>((m & 0xFFu) | ((m >> 16) & 0x100u));
      node->player = i == 0 ? root_player : static_cast<Player>((m >> 8) & 15u) - 1;
      node->prior = prior[i];
      node->explore_count = static_cast<int>(count[i]);
      node->total_reward = total[i];
      if ((m >> 20) & 1u) {  // outcome = Returns() of the proven / terminal position
        if (board) {
          const double v0 = static_cast<double>(static_cast<int>((m >> 21) & 3u) - 1);
          node->outcome = {v0, 0.0 - v0};  // 0.0 - 0.0 = +0.0: Returns() of a draw is {0, 0}, never -0
        } else if (count[i] > 0) {  // terminal node of a poker game: every visit added the same Returns()[player]
          node->outcome.assign(num_players, 0.0);
          if (node->player >= 0) node->outcome[node->player] = total[i] / count[i];
        }
      }
      const int nc = static_cast<int>(((m >> 12) & 0xFFu) | ((m >> 17) & 0x100u));
      node->children.resize(nc);
      for (int k = 0; k < nc; ++k) todo.push_back({&node->children[k], first[i] + static_cast<uint32_t>(k)});
    }
    return root;
  }

 private:
  osg_mcts_cfg Config() const {
    osg_mcts_cfg cfg{};
    cfg.uct_c = uct_c_;
    // wall-clock mode: the clock is the bound (mcts.cc:362-366); the device tree still needs a finite one for its
    // log table — 2^20 simulations per search, with the node budget of max_memory_mb collecting garbage as usual
    cfg.max_simulations = max_wall_clock_time_ > 0 ? (1 << 20) : max_simulations_;
    cfg.n_rollouts = rollout_ ? rollout_->n_rollouts() : 1;
    cfg.solve = solve_ ? 1 : 0;
    // max_nodes_ = (max_memory_mb << 20) / sizeof(SearchNode) + 1 (mcts.cc:214) with the reference's 88-byte
    // SearchNode (x86-64 libstdc++), so that the same max_memory_mb collects at the same tree size
    const int64_t nodes = (max_memory_mb_ << 20) / 88 + 1;
    cfg.max_nodes = static_cast<int32_t>(std::min<int64_t>(nodes, std::numeric_limits<int32_t>::max()));
    cfg.seed = static_cast<uint64_t>(seed_);
    cfg.index_offset = searches_;
    cfg.layout = 1;
    cfg.child_selection_policy = policy_ == ChildSelectionPolicy::PUCT ? 1 : 0;
    return cfg;
  }
  std::shared_ptr<Evaluator> evaluator_;
  RandomRolloutEvaluator* rollout_ = nullptr;
  double uct_c_;
  int max_simulations_;
  int64_t max_memory_mb_;
  bool solve_;
  int seed_;
  bool verbose_;
  int num_actions_, num_players_;
  ChildSelectionPolicy policy_;
  double dirichlet_alpha_, dirichlet_epsilon_;
  bool dont_return_chance_node_;
  double max_wall_clock_time_;
  std::mt19937 rng_;
  int64_t searches_ = 0;
  int64_t last_nodes_ = 0;
};

struct CFRInfoStateValues {  // cfr.h:42-98
  CFRInfoStateValues() {}
  CFRInfoStateValues(std::vector<Action> la, double init_value)
      : legal_actions(la), cumulative_regrets(la.size(), init_value), cumulative_policy(la.size(), init_value),
        current_policy(la.size(), 1.0 / la.size()) {}
  CFRInfoStateValues(std::vector<Action> la) : CFRInfoStateValues(la, 0) {}
  std::vector<Action> legal_actions;
  std::vector<double> cumulative_regrets, cumulative_policy, current_policy;
  bool empty() const { return legal_actions.empty(); }
  int num_actions() const { return static_cast<int>(legal_actions.size()); }
};
using CFRInfoStateValuesTable = std::unordered_map<std::string, CFRInfoStateValues>;  // cfr.h:103-104
using TabularPolicyTable = std::unordered_map<std::string, ActionsAndProbs>;

class DeviceTabularSolver {
 public:
  ~DeviceTabularSolver() { if (s_) osg_cfr_destroy(s_); }
  DeviceTabularSolver(const DeviceTabularSolver&) = delete;
  CFRInfoStateValuesTable InfoStateValuesTable() const {  // cfr.h:217
    Tables t = Download();
    CFRInfoStateValuesTable out;
    for (int i = 0; i < t.I; ++i) {
      CFRInfoStateValues v;
      for (int a = 0; a < t.nact[i]; ++a) {
        v.legal_actions.push_back(t.legal[i * t.A + a]);
        v.cumulative_regrets.push_back(t.regrets[i * t.A + a]);
        v.cumulative_policy.push_back(t.cum[i * t.A + a]);
        v.current_policy.push_back(t.cur[i * t.A + a]);
      }
      out.emplace(Key(i), std::move(v));
    }
    return out;
  }
  // infostate -> the player who acts there (the device tree's own record)
  std::unordered_map<std::string, Player> InfoStatePlayers() const {
    std::unordered_map<std::string, Player> out;
    const int I = static_cast<int>(sizes_[4]);
    for (int i = 0; i < I; ++i) out.emplace(Key(i), static_cast<Player>(osg_cfr_infostate_player(s_, i)));
    return out;
  }
  TabularPolicyTable TabularAveragePolicy() const { return PolicyTableOf(true); }  // cfr.h:205-211
  TabularPolicyTable TabularCurrentPolicy() const { return PolicyTableOf(false); }
  // cfr.h:198-211.  (The reference's policy objects are live views of the solver's table; these are snapshots
  // of the device tables at the time of the call.)
  inline std::shared_ptr<open_spiel::hip::Policy> AveragePolicy() const;
  inline std::shared_ptr<open_spiel::hip::Policy> CurrentPolicy() const;
  // Device policy evaluation: {expected returns [P], best-response values [P], NashConv, Exploitability}
  // of the tables' average (0) / current (1) policy, or of `table` (which == 2; keyed by infostate string).
  struct Evaluation {
    std::vector<double> expected_returns, best_response_values;
    double nash_conv = 0, exploitability = 0;
  };
  Evaluation EvaluatePolicy(int which, const TabularPolicyTable* table = nullptr) const {
    Evaluation ev;
    ev.expected_returns.resize(num_players_);
    ev.best_response_values.resize(num_players_);
    std::vector<double> dense;
    if (which == 2) {
      if (!table) SpielFatalError("EvaluatePolicy: no table");
      Tables t = Download();
      dense.assign(static_cast<size_t>(t.I) * t.A, 0.0);
      for (int i = 0; i < t.I; ++i) {
        auto it = table->find(Key(i));
        if (it == table->end()) SpielFatalError(Key(i) + " not found in policy.");
        for (int a = 0; a < t.nact[i]; ++a)
          for (const auto& ap : it->second)
            if (ap.first == t.legal[i * t.A + a]) dense[i * t.A + a] = ap.second;
      }
    }
    Check(osg_cfr_evaluate_policy(s_, which, dense.empty() ? nullptr : dense.data(), ev.expected_returns.data(),
                                  ev.best_response_values.data(), &ev.nash_conv, &ev.exploitability));
    return ev;
  }
  // Overwrites the device tables from a CFRInfoStateValuesTable (rows matched by infostate string).
  // allow_missing: infostates absent from `table` keep their current values (a reference MCCFR checkpoint
  // only lists the infostates its samples have reached).
  void LoadInfoStateValuesTable(const CFRInfoStateValuesTable& table, bool allow_missing = false) {
    Tables t = Download();
    for (int i = 0; i < t.I; ++i) {
      auto it = table.find(Key(i));
      if (it == table.end()) {
        if (allow_missing) continue;
        SpielFatalError(Key(i) + " missing from the table");
      }
      const CFRInfoStateValues& v = it->second;
      if (v.num_actions() != t.nact[i]) SpielFatalError(Key(i) + ": wrong number of actions");
      for (int a = 0; a < t.nact[i]; ++a) {
        t.regrets[i * t.A + a] = v.cumulative_regrets[a];
        t.cum[i * t.A + a] = v.cumulative_policy[a];
        t.cur[i * t.A + a] = v.current_policy[a];
      }
    }
    Check(osg_cfr_upload_tables(s_, t.regrets.data(), t.cum.data(), t.cur.data()));
  }
  int Iteration() const { return osg_cfr_iteration(s_); }
  int64_t NumInfoStates() const { return sizes_[4]; }
  int64_t NumHistories() const { return sizes_[0]; }
  osg_cfr* handle() const { return s_; }

 protected:
  // mccfr: 0 CFR family, 1 external sampling, 2 outcome sampling (with `epsilon`)
  DeviceTabularSolver(const Game& game, bool alternating, bool linear, bool rm_plus, int mccfr, double epsilon = 0.6,
                      bool random_initial_regrets = false, int seed = 0) {
    osg_cfr_cfg cfg{};
    cfg.random_initial_regrets = random_initial_regrets ? 1 : 0;  // the device's own counter streams (the reference: abseil)
    cfg.seed = static_cast<uint64_t>(seed);
    cfg.alternating_updates = alternating ? 1 : 0;
    cfg.linear_averaging = linear ? 1 : 0;
    cfg.regret_matching_plus = rm_plus ? 1 : 0;
    cfg.solver = mccfr;
    cfg.epsilon = epsilon;
    Check(osg_cfr_create(game.Ctx(), game.GameString().c_str(), &cfg, &s_));
    Check(osg_cfr_sizes(s_, sizes_));
    num_players_ = game.NumPlayers();
  }
  osg_cfr* s_ = nullptr;
  int64_t sizes_[6] = {0, 0, 0, 0, 0, 0};
  int num_players_ = 0;

 private:
  struct Tables {
    int I, A;
    std::vector<int32_t> nact, legal;
    std::vector<double> regrets, cum, cur, avg;
  };
  Tables Download() const {
    Tables t;
    t.I = static_cast<int>(sizes_[4]);
    t.A = static_cast<int>(sizes_[5]);
    const size_t n = static_cast<size_t>(t.I) * t.A;
    t.nact.resize(t.I); t.legal.resize(n); t.regrets.resize(n); t.cum.resize(n); t.cur.resize(n); t.avg.resize(n);
    Check(osg_cfr_tables(s_, t.nact.data(), t.legal.data(), t.regrets.data(), t.cum.data(), t.cur.data(), t.avg.data()));
    return t;
  }
  std::string Key(int i) const {
    char buf[512];
    if (osg_cfr_infostate_key(s_, i, buf, sizeof(buf)) < 0) SpielFatalError(osg_last_error());
    return buf;
  }
  // TabularBestResponse on the device: for every infostate its key, legal actions, acting player and the index of
  // the best-response action among the legal ones (each player responding to the others playing `table`);
  // values[p] = the responder p's best-response value at the root.
 public:
  std::vector<int32_t> BestResponseIndices(const TabularPolicyTable& table, std::vector<std::string>* keys,
                                           std::vector<std::vector<Action>>* legal, std::vector<int>* players,
                                           std::vector<double>* values) const {
    Tables t = Download();
    std::vector<double> dense(static_cast<size_t>(t.I) * t.A, 0.0);
    keys->clear(); legal->clear(); players->clear();
    for (int i = 0; i < t.I; ++i) {
      const std::string k = Key(i);
      auto it = table.find(k);
      if (it == table.end()) SpielFatalError(k + " not found in policy.");
      std::vector<Action> la;
      for (int a = 0; a < t.nact[i]; ++a) {
        la.push_back(t.legal[i * t.A + a]);
        for (const auto& ap : it->second)
          if (ap.first == t.legal[i * t.A + a]) dense[i * t.A + a] = ap.second;
      }
      keys->push_back(k);
      legal->push_back(std::move(la));
      players->push_back(osg_cfr_infostate_player(s_, i));
    }
    std::vector<int32_t> best(t.I);
    values->assign(num_players_, 0.0);
    Check(osg_cfr_best_response(s_, 2, dense.data(), best.data(), values->data()));
    return best;
  }
  // HistoryString() -> the responder's best-response value from that history on (TabularBestResponse::Value(history),
  // best_response.h:127-128), for every history of the game: one device pass (osg_cfr_best_response_history_values).
  // Infostates the table does not hold (it may leave out the responder's own) count as never played.
  std::unordered_map<std::string, double> BestResponseHistoryValues(const TabularPolicyTable& table, Player responder) const {
    Tables t = Download();
    std::vector<double> dense(static_cast<size_t>(t.I) * t.A, 0.0);
    for (int i = 0; i < t.I; ++i) {
      auto it = table.find(Key(i));
      if (it == table.end()) continue;
      for (int a = 0; a < t.nact[i]; ++a)
        for (const auto& ap : it->second)
          if (ap.first == t.legal[i * t.A + a]) dense[i * t.A + a] = ap.second;
    }
    const int64_t H = sizes_[0];
    std::vector<double> values(static_cast<size_t>(H));
    std::vector<int32_t> parent(static_cast<size_t>(H)), action(static_cast<size_t>(H));
    Check(osg_cfr_best_response_history_values(s_, 2, dense.data(), responder, values.data()));
    Check(osg_cfr_tree_edges(s_, parent.data(), action.data()));
    std::vector<std::string> text(static_cast<size_t>(H));
    std::unordered_map<std::string, double> out;
    for (int64_t h = 0; h < H; ++h) {  // parents come before their children (level order)
      if (parent[h] >= 0) text[h] = (parent[parent[h]] < 0 ? std::string() : text[parent[h]] + ", ") + std::to_string(action[h]);
      out[text[h]] = values[h];
    }
    return out;
  }

 protected:
  TabularPolicyTable PolicyTableOf(bool average) const {
    Tables t = Download();
    TabularPolicyTable out;
    for (int i = 0; i < t.I; ++i) {
      ActionsAndProbs ap;
      for (int a = 0; a < t.nact[i]; ++a)
        ap.emplace_back(t.legal[i * t.A + a], (average ? t.avg : t.cur)[i * t.A + a]);
      out.emplace(Key(i), std::move(ap));
    }
    return out;
  }
};

// cfr.h:33-39
constexpr const char* kSerializeMetaSectionHeader = "[Meta]";
constexpr const char* kSerializeGameSectionHeader = "[Game]";
constexpr const char* kSerializeSolverTypeSectionHeader = "[SolverType]";
constexpr const char* kSerializeSolverSpecificStateSectionHeader = "[SolverSpecificState]";
constexpr const char* kSerializeSolverValuesTableSectionHeader = "[SolverValuesTable]";
constexpr int kSerializationVersion = 1;  // spiel.h:45

inline std::string FormatDouble(double d, int double_precision) {  // utils/serialization.h:28-50
  if (double_precision == -1) {
    char buf[64];
    std::snprintf(buf, sizeof(buf), "%a", d);  // HexDoubleFormatter: lossless
    return buf;
  }
  std::stringstream stream;
  stream << std::fixed << std::setprecision(double_precision) << d;  // SimpleDoubleFormatter
  return stream.str();
}
// CFRInfoStateValues::Serialize (cfr.cc:509-532): "legal;regrets;cumulative policy;current policy".
inline std::string SerializeInfoStateValues(const CFRInfoStateValues& v, int double_precision) {
  std::string out;
  auto join = [&](const std::vector<double>& xs) {
    std::string r;
    for (size_t i = 0; i < xs.size(); ++i) r += (i ? "," : "") + FormatDouble(xs[i], double_precision);
    return r;
  };
  for (size_t i = 0; i < v.legal_actions.size(); ++i) out += (i ? "," : "") + std::to_string(v.legal_actions[i]);
  return out + ";" + join(v.cumulative_regrets) + ";" + join(v.cumulative_policy) + ";" + join(v.current_policy);
}
inline CFRInfoStateValues DeserializeInfoStateValues(const std::string& serialized) {  // cfr.cc:534-583
  CFRInfoStateValues res;
  std::vector<std::vector<std::string>> parts(1);
  std::string cur;
  for (char c : serialized) {
    if (c == ';') { parts.back().push_back(cur); cur.clear(); parts.emplace_back(); }
    else if (c == ',') { parts.back().push_back(cur); cur.clear(); }
    else cur.push_back(c);
  }
  parts.back().push_back(cur);
  if (parts.size() != 4) SpielFatalError("malformed CFRInfoStateValues: " + serialized);
  for (size_t i = 0; i < parts[0].size(); ++i) {
    res.legal_actions.push_back(std::strtoll(parts[0][i].c_str(), nullptr, 10));
    res.cumulative_regrets.push_back(std::strtod(parts[1].at(i).c_str(), nullptr));  // accepts "%a" hex floats
    res.cumulative_policy.push_back(std::strtod(parts[2].at(i).c_str(), nullptr));
    res.current_policy.push_back(std::strtod(parts[3].at(i).c_str(), nullptr));
  }
  return res;
}

class CFRSolverBase : public DeviceTabularSolver {  // cfr.h:188-304
 public:
  CFRSolverBase(const Game& game, bool alternating_updates, bool linear_averaging, bool regret_matching_plus,
                bool random_initial_regrets = false, int seed = 0)  // cfr.h:190-196
      : DeviceTabularSolver(game, alternating_updates, linear_averaging, regret_matching_plus, 0, 0.6, random_initial_regrets, seed),
        game_string_(game.Serialize()) {}
  virtual ~CFRSolverBase() = default;
  // The reference's text checkpoint (cfr.cc:284-307): [Meta] / [Game] / [SolverType] /
  // [SolverSpecificState] (the iteration) / [SolverValuesTable] rows "key<~>values<~>key<~>...".
  // double_precision == -1 writes lossless hex floats.
  std::string Serialize(int double_precision = -1, const std::string& delimiter = "<~>") const {
    if (double_precision < -1) SpielFatalError("double_precision must be >= -1");
    if (delimiter == "," || delimiter == ";")
      SpielFatalError("Please select a different delimiter,invalid values are \",\" and \";\".");
    std::string str = "# Automatically generated by OpenSpiel CFRSolverBase::Serialize\n";
    str += std::string(kSerializeMetaSectionHeader) + "\nVersion: " + std::to_string(kSerializationVersion) + "\n\n";
    str += std::string(kSerializeGameSectionHeader) + "\n" + game_string_ + "\n";
    str += std::string(kSerializeSolverTypeSectionHeader) + "\n" + SerializeThisType() + "\n";
    str += std::string(kSerializeSolverSpecificStateSectionHeader) + "\n" + std::to_string(Iteration()) + "\n";
    str += std::string(kSerializeSolverValuesTableSectionHeader) + "\n";
    bool first = true;
    for (const auto& kv : InfoStateValuesTable()) {  // cfr.cc:639-661
      if (kv.first.find(delimiter) != std::string::npos) SpielFatalError("Info state contains delimiter");
      if (!first) str += delimiter;
      first = false;
      str += kv.first + delimiter + SerializeInfoStateValues(kv.second, double_precision);
    }
    return str;
  }

 protected:
  virtual std::string SerializeThisType() const {  // cfr.h:261-263
    SpielFatalError("Serialization of the base class is not supported.");
  }

 private:
  std::string game_string_;

 public:
  void EvaluateAndUpdatePolicy() { Check(osg_cfr_iterate(s_, 1)); }  // cfr.cc:263-282
  void EvaluateAndUpdatePolicy(int iterations) { Check(osg_cfr_iterate(s_, iterations)); }  // one launch
};
class CFRSolver : public CFRSolverBase {  // cfr.h:310-330
 public:
  explicit CFRSolver(const Game& game) : CFRSolverBase(game, true, false, false) {}

 protected:
  std::string SerializeThisType() const override { return "CFRSolver"; }
};
class CFRPlusSolver : public CFRSolverBase {  // cfr.h:341-357
 public:
  explicit CFRPlusSolver(const Game& game) : CFRSolverBase(game, true, true, true) {}

 protected:
  std::string SerializeThisType() const override { return "CFRPlusSolver"; }
};
// cfr_br.h:34-56: every player minimises regret against the other players' best responses to the current
// policy (cfr_br.cc:48-83).  Best responses, the per-player passes and the regret matching all run on the device.
class CFRBRSolver : public CFRSolverBase {
 public:
  explicit CFRBRSolver(const Game& game) : CFRSolverBase(game, false, false, false) {}
  void EvaluateAndUpdatePolicy() { Check(osg_cfr_br_iterate(s_, 1)); }
  void EvaluateAndUpdatePolicy(int iterations) { Check(osg_cfr_br_iterate(s_, iterations)); }

 protected:
  std::string SerializeThisType() const override { return "CFRBRSolver"; }  // cfr_br.h:45
};

// PartiallyDeserializeCFRSolver + Deserialize{CFR,CFRPlus}Solver (cfr.cc:699-781).  Deviation: the
// reference rebuilds a CFRPlusSolver with linear_averaging = regret_matching_plus = false
// (cfr.h:349-353, an upstream slip); here the restored solver keeps CFR+ semantics.
template <class Solver>
inline std::unique_ptr<Solver> DeserializeSolver(const std::string& serialized, const std::string& expected_type,
                                                 const std::string& delimiter) {
  std::string sections[4];
  int current = -1;
  size_t pos = 0, table_at = std::string::npos;
  while (pos <= serialized.size()) {
    size_t nl = serialized.find('\n', pos);
    if (nl == std::string::npos) nl = serialized.size();
    const std::string line = serialized.substr(pos, nl - pos);
    pos = nl + 1;
    if (line.empty() || line[0] == '#') continue;
    if (line == kSerializeMetaSectionHeader) current = 0;
    else if (line == kSerializeGameSectionHeader) current = 1;
    else if (line == kSerializeSolverTypeSectionHeader) current = 2;
    else if (line == kSerializeSolverSpecificStateSectionHeader) current = 3;
    else if (line == kSerializeSolverValuesTableSectionHeader) { table_at = pos; break; }
    else if (current < 0) SpielFatalError("malformed solver checkpoint");
    else sections[current] += line;
  }
  if (table_at == std::string::npos) SpielFatalError("solver checkpoint without a values table");
  if (sections[2] != expected_type) SpielFatalError("checkpoint holds a " + sections[2] + ", not a " + expected_type);
  std::shared_ptr<const Game> game = LoadGame(sections[1]);
  auto solver = std::unique_ptr<Solver>(new Solver(*game));
  Check(osg_cfr_set_iteration(solver->handle(), std::stoi(sections[3])));
  CFRInfoStateValuesTable table;  // DeserializeCFRInfoStateValuesTable (cfr.cc:663-673)
  const std::string body = table_at <= serialized.size() ? serialized.substr(table_at) : std::string();
  std::vector<std::string> splits;
  for (size_t p = 0;;) {
    size_t q = body.find(delimiter, p);
    splits.push_back(body.substr(p, q == std::string::npos ? std::string::npos : q - p));
    if (q == std::string::npos) break;
    p = q + delimiter.size();
  }
  for (size_t i = 0; i + 1 < splits.size(); i += 2) table.emplace(splits[i], DeserializeInfoStateValues(splits[i + 1]));
  solver->LoadInfoStateValuesTable(table);
  return solver;
}
inline std::unique_ptr<CFRSolver> DeserializeCFRSolver(const std::string& serialized,
                                                       const std::string& delimiter = "<~>") {
  return DeserializeSolver<CFRSolver>(serialized, "CFRSolver", delimiter);
}
inline std::unique_ptr<CFRPlusSolver> DeserializeCFRPlusSolver(const std::string& serialized,
                                                               const std::string& delimiter = "<~>") {
  return DeserializeSolver<CFRPlusSolver>(serialized, "CFRPlusSolver", delimiter);
}
inline std::unique_ptr<CFRBRSolver> DeserializeCFRBRSolver(const std::string& serialized,
                                                           const std::string& delimiter = "<~>") {  // cfr_br.cc:85-96
  return DeserializeSolver<CFRBRSolver>(serialized, "CFRBRSolver", delimiter);
}

// algorithms::Exploitability / NashConv / ExpectedReturns of a tabular policy
// (tabular_exploitability.h, expected_returns.h), evaluated on the device.
inline double Exploitability(const Game& game, const TabularPolicyTable& policy) {
  return CFRSolver(game).EvaluatePolicy(2, &policy).exploitability;
}
inline double NashConv(const Game& game, const TabularPolicyTable& policy) {
  return CFRSolver(game).EvaluatePolicy(2, &policy).nash_conv;
}
inline std::vector<double> ExpectedReturns(const Game& game, const TabularPolicyTable& policy) {
  return CFRSolver(game).EvaluatePolicy(2, &policy).expected_returns;
}

}  // namespace algorithms

// ---- policy.h: Policy / TabularPolicy / UniformPolicy / PreferredActionPolicy and the policy factories ----------
class Policy {  // policy.h:69-150
 public:
  virtual ~Policy() = default;
  virtual ActionsAndProbs GetStatePolicy(const State& state) const { return GetStatePolicy(state, state.CurrentPlayer()); }
  virtual ActionsAndProbs GetStatePolicy(const State& state, Player player) const {
    return GetStatePolicy(state.InformationStateString(player));
  }
  virtual ActionsAndProbs GetStatePolicy(const std::string& /*info_state*/) const {
    SpielFatalError("GetStatePolicy(const std::string&) unimplemented.");
  }
  std::unordered_map<Action, double> GetStatePolicyAsMap(const State& state) const { return AsMap(GetStatePolicy(state)); }
  std::unordered_map<Action, double> GetStatePolicyAsMap(const std::string& info_state) const {
    return AsMap(GetStatePolicy(info_state));
  }
  std::pair<std::vector<Action>, std::vector<double>> GetStatePolicyAsParallelVectors(const State& state) const {
    return AsVectors(GetStatePolicy(state));
  }
  std::pair<std::vector<Action>, std::vector<double>> GetStatePolicyAsParallelVectors(const std::string& info_state) const {
    return AsVectors(GetStatePolicy(info_state));
  }

 private:
  static std::unordered_map<Action, double> AsMap(const ActionsAndProbs& ap) {
    std::unordered_map<Action, double> m;
    for (const auto& x : ap) m[x.first] = x.second;
    return m;
  }
  static std::pair<std::vector<Action>, std::vector<double>> AsVectors(const ActionsAndProbs& ap) {
    std::pair<std::vector<Action>, std::vector<double>> v;
    for (const auto& x : ap) { v.first.push_back(x.first); v.second.push_back(x.second); }
    return v;
  }
};

// The infostates of a game with their legal actions, from the device's flattened tree (get_all_states.cc /
// policy.cc:205-237 walk the game for the same list).
inline std::unordered_map<std::string, std::vector<Action>> AllInfoStates(const Game& game) {
  std::unordered_map<std::string, std::vector<Action>> out;
  for (const auto& kv : algorithms::CFRSolverBase(game, false, false, false).InfoStateValuesTable())
    out[kv.first] = kv.second.legal_actions;
  return out;
}

namespace algorithms {
// algorithms/get_legal_actions_map.h:32-33: infostate -> legal actions, for one player's infostates or (kInvalidPlayer)
// for everybody's; a depth limit walks the game on the host, without one the device's flattened tree has the answer
inline std::unordered_map<std::string, std::vector<Action>> GetLegalActionsMap(const Game& game, int depth_limit, Player player) {
  std::unordered_map<std::string, std::vector<Action>> out;
  if (depth_limit < 0) {
    CFRSolverBase solver(game, false, false, false);
    const auto who = solver.InfoStatePlayers();
    for (const auto& kv : solver.InfoStateValuesTable())
      if (player == kInvalidPlayer || who.at(kv.first) == player) out[kv.first] = kv.second.legal_actions;
    return out;
  }
  std::vector<std::pair<std::unique_ptr<State>, int>> todo;
  todo.emplace_back(game.NewInitialState(), 0);
  while (!todo.empty()) {
    std::unique_ptr<State> st = std::move(todo.back().first);
    const int depth = todo.back().second;
    todo.pop_back();
    if (st->IsTerminal() || depth > depth_limit) continue;
    if (!st->IsChanceNode() && (player == kInvalidPlayer || st->CurrentPlayer() == player))
      out[st->InformationStateString()] = st->LegalActions();
    for (Action a : st->LegalActions()) todo.emplace_back(st->Child(a), depth + 1);
  }
  return out;
}
}  // namespace algorithms

class TabularPolicy : public Policy {  // policy.h:158-283
 public:
  TabularPolicy() = default;
  explicit TabularPolicy(const Game& game) {  // the uniform random policy (policy.cc:205-237)
    for (const auto& kv : AllInfoStates(game))
      for (Action a : kv.second) policy_table_[kv.first].push_back({a, 1.0 / kv.second.size()});
  }
  TabularPolicy(algorithms::TabularPolicyTable table) : policy_table_(std::move(table)) {}  // (policy.h:165: from a table)
  TabularPolicy(const Game& game, const Policy& policy) {  // policy.h:167-171: tabularise any policy
    const auto all = AllInfoStates(game);
    bool by_key = true;
    try {
      if (!all.empty()) (void)policy.GetStatePolicy(all.begin()->first);
    } catch (const SpielException&) {
      by_key = false;  // the policy only answers for State objects: walk the game (policy.cc:205-237)
    }
    if (by_key) {
      for (const auto& kv : all) {
        ActionsAndProbs ap = policy.GetStatePolicy(kv.first);
        if (ap.empty()) SpielFatalError(kv.first + " not found in policy.");
        policy_table_[kv.first] = std::move(ap);
      }
      return;
    }
    std::vector<std::unique_ptr<State>> todo;
    todo.push_back(game.NewInitialState());
    while (!todo.empty()) {
      std::unique_ptr<State> st = std::move(todo.back());
      todo.pop_back();
      if (st->IsTerminal()) continue;
      if (st->IsChanceNode()) {
        for (const auto& ap : st->ChanceOutcomes()) todo.push_back(st->Child(ap.first));
        continue;
      }
      const std::string key = st->InformationStateString();
      if (!policy_table_.count(key)) policy_table_[key] = policy.GetStatePolicy(*st, st->CurrentPlayer());
      for (Action a : st->LegalActions()) todo.push_back(st->Child(a));
    }
  }
  using Policy::GetStatePolicy;
  ActionsAndProbs GetStatePolicy(const std::string& info_state) const override {  // policy.h:189-196
    auto it = policy_table_.find(info_state);
    return it == policy_table_.end() ? ActionsAndProbs{} : it->second;
  }
  void SetProb(const std::string& info_state, Action action, double prob) {  // policy.h:222-229
    for (auto& ap : policy_table_[info_state])
      if (ap.first == action) { ap.second = prob; return; }
    policy_table_[info_state].push_back({action, prob});
  }
  void SetStatePolicy(const std::string& info_state, const ActionsAndProbs& state_policy) {
    policy_table_[info_state] = state_policy;
  }
  algorithms::TabularPolicyTable& PolicyTable() { return policy_table_; }
  const algorithms::TabularPolicyTable& PolicyTable() const { return policy_table_; }
  int size() const { return static_cast<int>(policy_table_.size()); }
  std::string ToString() const { return ToStringSorted(); }  // (policy.cc:198-208 prints in hash-table order: any order)
  std::string ToStringSorted() const {  // policy.cc:210-229: "key:  action=prob action=prob", infostates sorted
    std::vector<std::string> keys;
    for (const auto& kv : policy_table_) keys.push_back(kv.first);
    std::sort(keys.begin(), keys.end());
    std::string str;
    for (const std::string& k : keys) {
      str += k + ": ";
      for (const auto& ap : policy_table_.at(k)) {
        std::ostringstream o;
        o << " " << ap.first << "=" << ap.second;
        str += o.str();
      }
      str += "\n";
    }
    return str;
  }

 private:
  algorithms::TabularPolicyTable policy_table_;
};

class UniformPolicy : public Policy {  // policy.h:318-334
 public:
  using Policy::GetStatePolicy;
  ActionsAndProbs GetStatePolicy(const State& state, Player player) const override {
    if (state.IsChanceNode()) return state.ChanceOutcomes();
    ActionsAndProbs ap;
    const std::vector<Action> legal = state.LegalActions(player);
    for (Action a : legal) ap.push_back({a, 1.0 / legal.size()});
    return ap;
  }
};

class PreferredActionPolicy : public Policy {  // policy.h:363-377, policy.cc:481-497
 public:
  explicit PreferredActionPolicy(std::vector<Action> preference_order) : order_(std::move(preference_order)) {}
  using Policy::GetStatePolicy;
  ActionsAndProbs GetStatePolicy(const State& state, Player player) const override {
    const std::vector<Action> legal = state.LegalActions(player);
    for (Action want : order_) {
      if (std::find(legal.begin(), legal.end(), want) == legal.end()) continue;
      ActionsAndProbs ap;
      for (Action a : legal) ap.push_back({a, a == want ? 1.0 : 0.0});
      return ap;
    }
    SpielFatalError("No preferred action found in the legal actions!");
  }

 private:
  std::vector<Action> order_;
};

inline TabularPolicy GetUniformPolicy(const Game& game) { return TabularPolicy(game); }  // policy.cc:239
inline TabularPolicy GetFirstActionPolicy(const Game& game) {  // policy.cc:375-398
  TabularPolicy p;
  for (const auto& kv : AllInfoStates(game)) {
    ActionsAndProbs ap;
    for (size_t i = 0; i < kv.second.size(); ++i) ap.push_back({kv.second[i], i == 0 ? 1.0 : 0.0});
    p.SetStatePolicy(kv.first, ap);
  }
  return p;
}
inline TabularPolicy GetEmptyTabularPolicy(const Game& game, bool initialize_to_uniform = false) {  // policy.cc:159-203
  if (initialize_to_uniform) return TabularPolicy(game);
  TabularPolicy p;
  for (const auto& kv : AllInfoStates(game)) {
    ActionsAndProbs ap;
    for (Action a : kv.second) ap.push_back({a, 0.0});
    p.SetStatePolicy(kv.first, ap);
  }
  return p;
}
inline TabularPolicy ToTabularPolicy(const Game& game, const Policy* policy) { return TabularPolicy(game, *policy); }

// The stock bots of spiel_bots.h:187-230 (what examples/mcts_example.cc and evaluate_bots_test.cc pit an MCTSBot
// against).  Host-only: they ask the position for its legal actions and draw on their own std::mt19937 (the reference
// draws on abseil's distributions over the same generator: same bots, another stream).
class UniformRandomBot : public Bot {
 public:
  UniformRandomBot(Player player_id, int seed) : player_id_(player_id), rng_(seed) {}
  void RestartAt(const State&) override {}
  Action Step(const State& state) override { return StepWithPolicy(state).second; }
  bool ProvidesPolicy() override { return true; }
  ActionsAndProbs GetPolicy(const State& state) override {
    ActionsAndProbs policy;
    const std::vector<Action> legal = state.LegalActions(player_id_);
    for (Action a : legal) policy.emplace_back(a, 1.0 / legal.size());
    return policy;
  }
  std::pair<ActionsAndProbs, Action> StepWithPolicy(const State& state) override {
    ActionsAndProbs policy = GetPolicy(state);
    if (policy.empty()) SpielFatalError("UniformRandomBot: no legal action for its player at this state");
    const int pick = std::uniform_int_distribution<int>(0, static_cast<int>(policy.size()) - 1)(rng_);
    return {policy, policy[pick].first};
  }
  bool IsClonable() const override { return true; }
  std::unique_ptr<Bot> Clone() override { return std::make_unique<UniformRandomBot>(*this); }

 protected:
  const Player player_id_;
  std::mt19937 rng_;
};
// A UniformRandomBot that follows the game in a state of its own: it exists to check that the run loop tells every
// bot every action (spiel_bots.cc: StatefulRandomBot).
class StatefulRandomBot : public UniformRandomBot {
 public:
  StatefulRandomBot(const Game& game, Player player_id, int seed)
      : UniformRandomBot(player_id, seed), state_(game.NewInitialState()) {}
  StatefulRandomBot(const StatefulRandomBot& other) : UniformRandomBot(other), state_(other.state_->Clone()) {}
  void Restart() override { state_ = state_->GetGame()->NewInitialState(); }
  void RestartAt(const State& state) override { state_ = state.Clone(); }
  void InformAction(const State& state, Player, Action action) override {
    CheckSame(state);
    state_->ApplyAction(action);
  }
  ActionsAndProbs GetPolicy(const State& state) override {
    CheckSame(state);
    return UniformRandomBot::GetPolicy(*state_);
  }
  std::pair<ActionsAndProbs, Action> StepWithPolicy(const State&) override {
    std::pair<ActionsAndProbs, Action> out = UniformRandomBot::StepWithPolicy(*state_);
    state_->ApplyAction(out.second);
    return out;
  }
  std::unique_ptr<Bot> Clone() override { return std::make_unique<StatefulRandomBot>(*this); }

 private:
  void CheckSame(const State& other) const {
    if (other.History() != state_->History() || other.CurrentPlayer() != state_->CurrentPlayer() ||
        other.LegalActions() != state_->LegalActions())
      SpielFatalError("StatefulRandomBot: the run loop's state and the bot's own have diverged");
    if (!other.IsChanceNode() && other.ObservationTensor(other.CurrentPlayer()) != state_->ObservationTensor(state_->CurrentPlayer()))
      SpielFatalError("StatefulRandomBot: observation tensors differ");
  }
  std::unique_ptr<State> state_;
};
// Samples its action from a Policy (spiel_bots.cc: PolicyBot).
class PolicyBot : public Bot {
 public:
  PolicyBot(int seed, std::shared_ptr<Policy> policy) : rng_(seed), policy_(std::move(policy)) {}
  void RestartAt(const State&) override {}
  Action Step(const State& state) override { return StepWithPolicy(state).second; }
  bool ProvidesPolicy() override { return true; }
  ActionsAndProbs GetPolicy(const State& state) override { return policy_->GetStatePolicy(state); }
  std::pair<ActionsAndProbs, Action> StepWithPolicy(const State& state) override {
    ActionsAndProbs policy = GetPolicy(state);
    if (policy.empty()) SpielFatalError("PolicyBot: the policy has no entry for this state");
    return {policy, SampleAction(policy, rng_).first};
  }
  bool IsClonable() const override { return true; }
  std::unique_ptr<Bot> Clone() override { return std::make_unique<PolicyBot>(*this); }

 private:
  std::mt19937 rng_;
  std::shared_ptr<Policy> policy_;
};
// Plays the first legal action of its preference list (spiel_bots.cc: FixedActionPreferenceBot).
class FixedActionPreferenceBot : public Bot {
 public:
  FixedActionPreferenceBot(Player player_id, const std::vector<Action>& actions) : player_id_(player_id), actions_(actions) {}
  void RestartAt(const State&) override {}
  Action Step(const State& state) override { return StepWithPolicy(state).second; }
  bool ProvidesPolicy() override { return true; }
  ActionsAndProbs GetPolicy(const State& state) override {
    const std::vector<Action> legal = state.LegalActions(player_id_);
    for (Action a : actions_)
      if (std::find(legal.begin(), legal.end(), a) != legal.end()) return {{a, 1.0}};
    SpielFatalError("No legal actions found in preferred list");
  }
  std::pair<ActionsAndProbs, Action> StepWithPolicy(const State& state) override {
    ActionsAndProbs policy = GetPolicy(state);
    return {policy, policy[0].first};
  }
  bool IsClonable() const override { return true; }
  std::unique_ptr<Bot> Clone() override { return std::make_unique<FixedActionPreferenceBot>(*this); }

 private:
  const Player player_id_;
  std::vector<Action> actions_;
};
inline std::unique_ptr<Bot> MakeUniformRandomBot(Player player_id, int seed) { return std::make_unique<UniformRandomBot>(player_id, seed); }
inline std::unique_ptr<Bot> MakeStatefulRandomBot(const Game& game, Player player_id, int seed) {
  return std::make_unique<StatefulRandomBot>(game, player_id, seed);
}
inline std::unique_ptr<Bot> MakePolicyBot(int seed, std::shared_ptr<Policy> policy) { return std::make_unique<PolicyBot>(seed, std::move(policy)); }
inline std::unique_ptr<Bot> MakePolicyBot(const Game&, Player, int seed, std::shared_ptr<Policy> policy) {
  return MakePolicyBot(seed, std::move(policy));  // (the reference ignores game and player as well)
}
inline std::unique_ptr<Bot> MakeFixedActionPreferenceBot(Player player_id, const std::vector<Action>& actions) {
  return std::make_unique<FixedActionPreferenceBot>(player_id, actions);
}
inline std::shared_ptr<Policy> algorithms::DeviceTabularSolver::AveragePolicy() const {
  return std::make_shared<TabularPolicy>(TabularAveragePolicy());
}
inline std::shared_ptr<Policy> algorithms::DeviceTabularSolver::CurrentPolicy() const {
  return std::make_shared<TabularPolicy>(TabularCurrentPolicy());
}

namespace algorithms {

// best_response.h:38-130.  The best response of `best_responder` to `policy` (the other players follow it),
// computed on the device over the flattened tree (k_policy_eval): Value at the root, the deterministic
// best-response policy / actions for every infostate of the responder (ties and unreachable infostates: the
// lowest action, best_response.cc:194-227).
class TabularBestResponse {
 public:
  TabularBestResponse(const Game& game, Player best_responder, const Policy* policy)
      : game_(LoadGame(game.ToString())), solver_(game, false, false, false), responder_(best_responder),
        num_players_(game.NumPlayers()) {
    if (best_responder < 0 || best_responder >= num_players_) SpielFatalError("TabularBestResponse: no such player");
    SetPolicy(policy);
  }
  TabularBestResponse(const Game& game, Player best_responder, const TabularPolicyTable& table)
      : game_(LoadGame(game.ToString())), solver_(game, false, false, false), responder_(best_responder),
        num_players_(game.NumPlayers()), table_(table) {
    if (best_responder < 0 || best_responder >= num_players_) SpielFatalError("TabularBestResponse: no such player");
  }
  void SetPolicy(const Policy* policy) {  // best_response.h:132-150
    if (!policy) SpielFatalError("TabularBestResponse: null policy");
    table_ = TabularPolicy(*game_, *policy).PolicyTable();
    computed_ = false;
    history_values_.clear();
  }
  void SetPolicy(const TabularPolicyTable& table) {
    table_ = table;
    computed_ = false;
    history_values_.clear();
  }
  double Value() { Compute(); return value_; }                               // Value(*root) (best_response.h:127-128)
  double Value(const std::string& history) {  // best_response.h:127-128: history = State::HistoryString()
    if (history.empty()) return Value();
    if (history_values_.empty()) history_values_ = solver_.BestResponseHistoryValues(table_, responder_);
    auto it = history_values_.find(history);
    if (it == history_values_.end()) SpielFatalError("TabularBestResponse::Value: no such history: " + history);
    return it->second;
  }
  double Value(const State& state) { return Value(state.HistoryString()); }
  std::unordered_map<std::string, Action> GetBestResponseActions() { Compute(); return actions_; }
  TabularPolicy GetBestResponsePolicy() {
    Compute();
    TabularPolicy p;
    for (const auto& kv : actions_) {
      ActionsAndProbs ap;
      for (Action a : legal_.at(kv.first)) ap.push_back({a, a == kv.second ? 1.0 : 0.0});
      p.SetStatePolicy(kv.first, ap);
    }
    return p;
  }

 private:
  void Compute() {
    if (computed_) return;
    std::vector<std::string> keys;
    std::vector<std::vector<Action>> legal;
    std::vector<int> players;
    std::vector<int32_t> best = solver_.BestResponseIndices(table_, &keys, &legal, &players, &values_);
    actions_.clear();
    legal_.clear();
    for (size_t i = 0; i < keys.size(); ++i) {
      if (players[i] != responder_) continue;
      actions_[keys[i]] = legal[i][best[i]];
      legal_[keys[i]] = legal[i];
    }
    value_ = values_[responder_];
    computed_ = true;
  }
  std::shared_ptr<const Game> game_;
  CFRSolverBase solver_;
  Player responder_;
  int num_players_;
  TabularPolicyTable table_;
  bool computed_ = false;
  double value_ = 0;
  std::vector<double> values_;
  std::unordered_map<std::string, Action> actions_;
  std::unordered_map<std::string, std::vector<Action>> legal_;
  std::unordered_map<std::string, double> history_values_;
};

// algorithms::Exploitability / NashConv / ExpectedReturns of ANY Policy (tabular_exploitability.h:30-60,
// expected_returns.h): tabularised over the game's infostates, judged on the device.
inline double Exploitability(const Game& game, const Policy& policy) {
  return Exploitability(game, TabularPolicy(game, policy).PolicyTable());
}
// (use_state_get_policy — tabular_exploitability.h:52-60 — chooses which GetStatePolicy overload the reference
// calls; the tabularisation here tries the infostate-string one and falls back to the State one by itself)
inline double NashConv(const Game& game, const Policy& policy, bool /*use_state_get_policy*/ = false) {
  return NashConv(game, TabularPolicy(game, policy).PolicyTable());
}
inline std::vector<double> ExpectedReturns(const Game& game, const Policy& policy) {
  return ExpectedReturns(game, TabularPolicy(game, policy).PolicyTable());
}
// expected_returns.h:47-51 for the case every caller on this path uses: the whole game from its root
inline std::vector<double> ExpectedReturns(const State& state, const Policy& joint_policy, int depth_limit) {
  if (state.MoveNumber() != 0 || depth_limit >= 0)
    SpielFatalError("ExpectedReturns: the device pass evaluates the whole game from its initial state");
  return ExpectedReturns(*state.GetGame(), joint_policy);
}

}  // namespace algorithms

// policy.cc:400-430 GetPrefActionPolicy: at every infostate the first action of `pref_actions` that is legal, with
// probability 1.
inline TabularPolicy GetPrefActionPolicy(const Game& game, const std::vector<Action>& pref_actions) {
  TabularPolicy p;
  for (const auto& kv : AllInfoStates(game)) {
    Action chosen = kInvalidAction;
    for (Action want : pref_actions)
      if (std::find(kv.second.begin(), kv.second.end(), want) != kv.second.end()) { chosen = want; break; }
    if (chosen == kInvalidAction) SpielFatalError("GetPrefActionPolicy: no preferred action is legal at " + kv.first);
    ActionsAndProbs ap;
    for (Action a : kv.second) ap.push_back({a, a == chosen ? 1.0 : 0.0});
    p.SetStatePolicy(kv.first, ap);
  }
  return p;
}

// kuhn_poker.h:40-47, kuhn_poker.cc:439-474
namespace kuhn_poker {
enum ActionType { kPass = 0, kBet = 1 };
inline constexpr int kNumInfoStatesP0 = 6, kNumInfoStatesP1 = 6;  // kuhn_poker.h:42-43 (two players)
inline TabularPolicy GetAlwaysPassPolicy(const Game& game) { return GetPrefActionPolicy(game, {ActionType::kPass}); }
inline TabularPolicy GetAlwaysBetPolicy(const Game& game) { return GetPrefActionPolicy(game, {ActionType::kBet}); }
// the alpha-family of Nash equilibria of 2-player Kuhn poker, alpha in [0, 1/3]; its value for player 0 is -1/18
inline TabularPolicy GetOptimalPolicy(double alpha) {
  if (!(alpha >= 0.0 && alpha <= 1.0 / 3)) SpielFatalError("GetOptimalPolicy: alpha must lie in [0, 1/3]");
  const double three_alpha = 3 * alpha;
  algorithms::TabularPolicyTable policy;
  // every infostate has two actions: Pass (0) and Bet (1)
  policy["0"] = {{0, 1 - alpha}, {1, alpha}};            // player 0
  policy["0pb"] = {{0, 1}, {1, 0}};
  policy["1"] = {{0, 1}, {1, 0}};
  policy["1pb"] = {{0, 2. / 3. - alpha}, {1, 1. / 3. + alpha}};
  policy["2"] = {{0, 1 - three_alpha}, {1, three_alpha}};
  policy["2pb"] = {{0, 0}, {1, 1}};
  policy["0p"] = {{0, 2. / 3.}, {1, 1. / 3.}};           // player 1
  policy["0b"] = {{0, 1}, {1, 0}};
  policy["1p"] = {{0, 1}, {1, 0}};
  policy["1b"] = {{0, 2. / 3.}, {1, 1. / 3.}};
  policy["2p"] = {{0, 0}, {1, 1}};
  policy["2b"] = {{0, 0}, {1, 1}};
  return TabularPolicy(policy);
}
}  // namespace kuhn_poker

// leduc_poker.h:64, leduc_poker.cc:872-888
namespace leduc_poker {
enum ActionType { kFold = 0, kCall = 1, kRaise = 2 };
inline constexpr int kNumInfoStates = 936;  // leduc_poker.h:72 (two players)
inline TabularPolicy GetAlwaysFoldPolicy(const Game& game) { return GetPrefActionPolicy(game, {ActionType::kFold, ActionType::kCall}); }
inline TabularPolicy GetAlwaysCallPolicy(const Game& game) { return GetPrefActionPolicy(game, {ActionType::kCall}); }
inline TabularPolicy GetAlwaysRaisePolicy(const Game& game) { return GetPrefActionPolicy(game, {ActionType::kRaise, ActionType::kCall}); }
}  // namespace leduc_poker

namespace algorithms {

// algorithms/get_all_states.h:30-48: every state of the game reachable from its root, keyed by ToString(); the walk
// stops where a key repeats (stop_at_duplicates) — a host-side walk over one-state batches, for small games.
inline std::map<std::string, std::unique_ptr<State>> GetAllStates(const Game& game, int depth_limit, bool include_terminals,
                                                                  bool include_chance_states, bool stop_at_duplicates = true) {
  std::map<std::string, std::unique_ptr<State>> all;
  std::vector<std::pair<std::unique_ptr<State>, int>> todo;
  todo.emplace_back(game.NewInitialState(), 0);
  while (!todo.empty()) {
    std::unique_ptr<State> st = std::move(todo.back().first);
    const int depth = todo.back().second;
    todo.pop_back();
    const bool terminal = st->IsTerminal(), chance = !terminal && st->IsChanceNode();
    const bool include = terminal ? include_terminals : (chance ? include_chance_states : true);
    bool seen = false;
    if (include) {
      const std::string key = st->ToString();
      seen = all.count(key) > 0;
      if (!seen) all[key] = st->Clone();
    }
    if (terminal || (depth_limit >= 0 && depth >= depth_limit) || (seen && stop_at_duplicates)) continue;
    for (Action a : st->LegalActions()) todo.emplace_back(st->Child(a), depth + 1);
  }
  return all;
}

// algorithms/get_all_histories.h:40-42: one State per history, in depth-first pre-order (children in action order);
// terminals and chance nodes on request; depth_limit < 0: the whole game.  A host-side walk over one-state batches.
inline std::vector<std::unique_ptr<State>> GetAllHistories(const Game& game, int depth_limit = -1, bool include_terminals = false,
                                                           bool include_chance_states = false) {
  std::vector<std::unique_ptr<State>> all;
  std::vector<std::pair<std::unique_ptr<State>, int>> todo;
  todo.emplace_back(game.NewInitialState(), 0);
  while (!todo.empty()) {
    std::unique_ptr<State> st = std::move(todo.back().first);
    const int depth = todo.back().second;
    todo.pop_back();
    if (st->IsTerminal()) {
      if (include_terminals) all.push_back(std::move(st));
      continue;
    }
    if (depth_limit >= 0 && depth > depth_limit) continue;
    const std::vector<Action> legal = st->LegalActions();
    for (auto it = legal.rbegin(); it != legal.rend(); ++it) todo.emplace_back(st->Child(*it), depth + 1);  // (popped in action order)
    if (!st->IsChanceNode() || include_chance_states) all.push_back(std::move(st));
  }
  if (all.empty()) SpielFatalError("GetSubgameHistories returned 0 histories!");
  return all;
}

enum class AverageType { kSimple, kFull };

// The MCCFR solvers' text checkpoints (external_sampling_mccfr.cc:82-120,233-288;
// outcome_sampling_mccfr.cc:76-112,243-300): same sections in the same order as the reference.
// [SolverRNG] holds the solver's std::mt19937 exactly as the reference writes it (operator<<), so checkpoints travel
// both ways, followed by one line of this engine's own, "counter <seed> <next trajectory>" — the position of the
// counter streams its mini-batches draw from — which the reference's reader (operator>>) never reaches.
constexpr const char* kSerializeSolverRNGSectionHeader = "[SolverRNG]";
constexpr const char* kSerializeSolverAverageTypeSectionHeader = "[SolverAverageType]";
constexpr const char* kSerializeSolverEpsilonSectionHeader = "[SolverEpsilon]";
constexpr const char* kSerializeSolverDefaultPolicySectionHeader = "[SolverDefaultPolicy]";

inline std::string SerializeValuesTable(const CFRInfoStateValuesTable& table, int double_precision,
                                        const std::string& delimiter) {  // cfr.cc:639-661
  std::string str;
  bool first = true;
  for (const auto& kv : table) {
    if (kv.first.find(delimiter) != std::string::npos) SpielFatalError("Info state contains delimiter");
    if (!first) str += delimiter;
    first = false;
    str += kv.first + delimiter + SerializeInfoStateValues(kv.second, double_precision);
  }
  return str;
}
inline CFRInfoStateValuesTable DeserializeValuesTable(const std::string& body, const std::string& delimiter) {
  CFRInfoStateValuesTable table;  // cfr.cc:663-673
  std::vector<std::string> splits;
  for (size_t p = 0;;) {
    size_t q = body.find(delimiter, p);
    splits.push_back(body.substr(p, q == std::string::npos ? std::string::npos : q - p));
    if (q == std::string::npos) break;
    p = q + delimiter.size();
  }
  for (size_t i = 0; i + 1 < splits.size(); i += 2) table.emplace(splits[i], DeserializeInfoStateValues(splits[i + 1]));
  return table;
}
// cfr.h:111-119 / cfr.cc:639-673, with the reference's signatures
inline void SerializeCFRInfoStateValuesTable(const CFRInfoStateValuesTable& info_states, std::string* result,
                                             int double_precision, std::string delimiter = "<~>") {
  if (delimiter == "," || delimiter == ";")
    SpielFatalError("Please select a different delimiter,invalid values are \",\" and \";\".");
  if (info_states.empty()) return;
  *result += SerializeValuesTable(info_states, double_precision, delimiter);
}
inline void DeserializeCFRInfoStateValuesTable(const std::string& serialized, CFRInfoStateValuesTable* result,
                                               std::string delimiter = "<~>") {
  if (serialized.empty()) return;
  for (auto& kv : DeserializeValuesTable(serialized, delimiter)) result->insert(std::move(kv));
}
// Splits a checkpoint into its top-level sections; [SolverSpecificState] keeps its line structure.
struct PartialCheckpoint {
  std::string game, solver_type, table;
  std::vector<std::string> specific;  // lines of [SolverSpecificState]
};
inline PartialCheckpoint PartiallyDeserializeSolver(const std::string& serialized) {  // cfr.cc:699-756
  PartialCheckpoint out;
  int current = -1;
  size_t pos = 0;
  bool have_table = false;
  while (pos <= serialized.size()) {
    size_t nl = serialized.find('\n', pos);
    if (nl == std::string::npos) nl = serialized.size();
    const std::string line = serialized.substr(pos, nl - pos);
    pos = nl + 1;
    if (line.empty() || line[0] == '#') continue;
    if (line == kSerializeMetaSectionHeader) current = 0;
    else if (line == kSerializeGameSectionHeader) current = 1;
    else if (line == kSerializeSolverTypeSectionHeader) current = 2;
    else if (line == kSerializeSolverSpecificStateSectionHeader) current = 3;
    else if (line == kSerializeSolverValuesTableSectionHeader) {
      out.table = pos <= serialized.size() ? serialized.substr(pos) : std::string();
      have_table = true;
      break;
    } else if (current == 1) out.game += line;
    else if (current == 2) out.solver_type += line;
    else if (current == 3) out.specific.push_back(line);
    else if (current < 0) SpielFatalError("malformed solver checkpoint");
  }
  if (!have_table) SpielFatalError("solver checkpoint without a values table");
  return out;
}

class ExternalSamplingMCCFRSolver : public DeviceTabularSolver {  // external_sampling_mccfr.h:57-113
 public:
  explicit ExternalSamplingMCCFRSolver(const Game& game, int seed = 0, AverageType avg_type = AverageType::kSimple)
      : DeviceTabularSolver(game, true, false, false, 1), seed_(seed), rng_(seed), avg_type_(avg_type),
        game_string_(game.Serialize()) {
    Check(osg_mccfr_set_average_type(s_, avg_type == AverageType::kFull ? 1 : 0));
  }
  // One UpdateRegrets per player, each seeing the previous one's update, then — AverageType::kFull — one
  // FullUpdateAverage (:71-80), drawing from the solver's own std::mt19937(seed) as the reference's does (:66):
  // a solver seeded like the reference's follows it iteration by iteration.
  void RunIteration() { RunIteration(&rng_); }
  // The same iteration with every draw taken from the caller's generator exactly as the reference takes it
  // (dist_(*rng), a std::uniform_real_distribution<double>, once per chance node and once per opponent node
  // in visiting order; external_sampling_mccfr.h:63-100, .cc:122-154): seeded alike, the tables follow the
  // reference's iteration by iteration.  The traversal runs on the device over a prefix of the generator's
  // sequence; the generator is then advanced by exactly the draws the traversal used.
  void RunIteration(std::mt19937* rng) {
    std::uniform_real_distribution<double> dist(0.0, 1.0);
    for (int p = 0; p < num_players_; ++p) {
      std::mt19937 ahead = *rng;
      const int want = static_cast<int>(std::min<int64_t>(sizes_[0], 1 << 20));  // a traversal visits <= H nodes
      uniforms_.resize(static_cast<size_t>(want));
      for (double& u : uniforms_) u = dist(ahead);
      int32_t used = 0;
      Check(osg_mccfr_sample_uniforms(s_, p, uniforms_.data(), want, &used));
      Check(osg_mccfr_apply_deltas(s_));
      for (int32_t k = 0; k < used; ++k) (void)dist(*rng);
    }
    if (avg_type_ == AverageType::kFull) Check(osg_mccfr_full_average(s_, 1.0));
  }
  // `trajectories` traverser passes as ONE mini-batch against the current tables (kFull: followed by
  // FullUpdateAverage weighted by the trajectories / players iterations the batch stands for).
  void RunMiniBatch(int64_t trajectories) {
    Check(osg_mccfr_iterate(s_, seed_, next_, trajectories));
    next_ += trajectories;
  }
  std::string Serialize(int double_precision = -1, const std::string& delimiter = "<~>") const {  // :82-120
    if (double_precision < -1) SpielFatalError("double_precision must be >= -1");
    std::string str = "# Automatically generated by OpenSpiel ExternalSamplingMCCFRSolver::Serialize\n";
    str += std::string(kSerializeMetaSectionHeader) + "\nVersion: " + std::to_string(kSerializationVersion) + "\n\n";
    str += std::string(kSerializeGameSectionHeader) + "\n" + game_string_ + "\n";
    str += std::string(kSerializeSolverTypeSectionHeader) + "\nExternalSamplingMCCFRSolver\n";
    str += std::string(kSerializeSolverSpecificStateSectionHeader) + "\n";
    // [SolverRNG]: the generator's state as the reference writes it (operator<< of std::mt19937, :100-102), then —
    // this engine's addition, ignored by the reference's reader (operator>> stops after the state) — the
    // position of the counter streams the mini-batches draw from.
    std::ostringstream rng_stream;
    rng_stream << rng_;
    str += std::string(kSerializeSolverRNGSectionHeader) + "\n" + rng_stream.str() + "\ncounter " +
           std::to_string(seed_) + " " + std::to_string(next_) + "\n";
    str += std::string(kSerializeSolverAverageTypeSectionHeader) +
           (avg_type_ == AverageType::kFull ? "\nFullAverageType\n" : "\nSimpleAverageType\n");
    str += std::string(kSerializeSolverDefaultPolicySectionHeader) + "\nUniformPolicy:\n";  // policy.h:330-333
    str += std::string(kSerializeSolverValuesTableSectionHeader) + "\n";
    return str + SerializeValuesTable(InfoStateValuesTable(), double_precision, delimiter);
  }
  // The same mini-batch with its trajectories sharded over the ranks of `comm`: each rank samples its
  // slice of the global index range, ONE all-reduce(sum) of the two adjacent [I, Amax] delta tables over
  // xGMI, then every rank folds identical deltas in — all ranks end with identical tables, and with the
  // tables a single GPU would have computed up to fp64 summation order.
  void RunShardedMiniBatch(int64_t trajectories, Communicator& comm, bool overlap = false) {
    const auto [first, count] = comm.Shard(trajectories);
    const int64_t n = sizes_[4] * sizes_[5];
    if (overlap) {
      // Double-buffered: mini-batch k's deltas are summed over the ranks on the communicator's stream while
      // mini-batch k + 1 is sampled (against tables without k's deltas: stale by one mini-batch); k's deltas are
      // folded when they have arrived, before mini-batch k + 2.  FinishSharded() folds what is still in flight.
      double* buf = nullptr;
      Check(osg_mccfr_spare_delta_buffer(s_, static_cast<int>(minibatches_ & 1), &buf));
      Check(osg_mccfr_sample_into(s_, seed_, next_ + first, count, buf));
      FinishSharded(comm);  // mini-batch k - 1: its all-reduce has been running under the sampling of k
      if (comm.world() > 1) comm.BeginAllReduceSum(buf, 2 * n);
      pending_ = buf;
      ++minibatches_;
      next_ += trajectories;
      return;
    }
    Check(osg_mccfr_sample(s_, seed_, next_ + first, count));
    if (comm.world() > 1) {
      double *dreg = nullptr, *dpol = nullptr;
      Check(osg_mccfr_delta_ptrs(s_, &dreg, &dpol));
      if (dpol == dreg + n) {
        comm.AllReduceSum(dreg, 2 * n);
      } else {
        comm.AllReduceSum(dreg, n);
        comm.AllReduceSum(dpol, n);
      }
    }
    Check(osg_mccfr_apply_deltas(s_));
    next_ += trajectories;
  }
  // Folds the deltas of the last overlapped mini-batch (a no-op when nothing is pending).
  void FinishSharded(Communicator& comm) {
    if (!pending_) return;
    if (comm.world() > 1) comm.EndAllReduce();
    Check(osg_mccfr_apply_deltas_from(s_, pending_));
    pending_ = nullptr;
  }
  void RestoreCounter(uint64_t seed, int64_t next) { seed_ = seed; next_ = next; }
  void RestoreGenerator(const std::string& state) {  // external_sampling_mccfr.cc:262-264
    std::istringstream in(state);
    in >> rng_;
    if (in.fail()) SpielFatalError("[SolverRNG] does not hold a std::mt19937 state");
  }
  int64_t TrajectoriesRun() const { return next_; }
  AverageType average_type() const { return avg_type_; }

 private:
  uint64_t seed_;
  int64_t next_ = 0;
  std::mt19937 rng_;
  AverageType avg_type_;
  std::string game_string_;
  std::vector<double> uniforms_;
  double* pending_ = nullptr;  // RunShardedMiniBatch(overlap): the delta buffer whose all-reduce is in flight
  int64_t minibatches_ = 0;
};

class OutcomeSamplingMCCFRSolver : public DeviceTabularSolver {  // outcome_sampling_mccfr.h:40-107
 public:
  static constexpr double kDefaultEpsilon = 0.6;
  explicit OutcomeSamplingMCCFRSolver(const Game& game, double epsilon = kDefaultEpsilon, int seed = -1)
      : DeviceTabularSolver(game, true, false, false, 2, epsilon), seed_(seed < 0 ? 0 : seed),
        rng_(seed >= 0 ? static_cast<std::mt19937::result_type>(seed) : std::mt19937::default_seed),  // outcome_sampling_mccfr.cc:44-48
        epsilon_(epsilon), game_string_(game.Serialize()) {}
  // One SampleEpisode per player, each seeing the previous one's update (:67-74), drawing on the solver's own
  // std::mt19937 like the reference's RunIteration() (outcome_sampling_mccfr.h:63): the generator advances with the
  // iterations and travels in the checkpoint.
  void RunIteration() { RunIteration(&rng_); }
  // outcome_sampling_mccfr.h:61-62.  The reference draws through abseil's distributions (unspecified streams);
  // here every episode takes a fresh 64-bit stream key from the caller's generator: the same distribution of
  // episodes, reproducible from the generator's state, not the reference's draw sequence.
  void RunIteration(std::mt19937* rng) {
    for (int p = 0; p < num_players_; ++p) {
      const uint64_t key = (static_cast<uint64_t>((*rng)()) << 32) | (*rng)();
      Check(osg_mccfr_iterate(s_, key, next_++, 1));
    }
  }
  void RunMiniBatch(int64_t episodes) {  // `episodes` sampled paths as ONE mini-batch
    Check(osg_mccfr_iterate(s_, seed_, next_, episodes));
    next_ += episodes;
  }
  std::string Serialize(int double_precision = -1, const std::string& delimiter = "<~>") const {  // :76-112
    if (double_precision < -1) SpielFatalError("double_precision must be >= -1");
    std::string str = "# Automatically generated by OpenSpiel OutcomeSamplingMCCFRSolver::Serialize\n";
    str += std::string(kSerializeMetaSectionHeader) + "\nVersion: " + std::to_string(kSerializationVersion) + "\n\n";
    str += std::string(kSerializeGameSectionHeader) + "\n" + game_string_ + "\n";
    str += std::string(kSerializeSolverTypeSectionHeader) + "\nOutcomeSamplingMCCFRSolver\n";
    str += std::string(kSerializeSolverSpecificStateSectionHeader) + "\n";
    // [SolverRNG]: the generator as the reference writes it (operator<< of std::mt19937, :94-98), then this engine's
    // counter-stream position on a line of its own (the reference's `rng_stream >> rng_` stops before it)
    std::ostringstream rng_stream;
    rng_stream << rng_;
    str += std::string(kSerializeSolverRNGSectionHeader) + "\n" + rng_stream.str() + "\ncounter " + std::to_string(seed_) + " " +
           std::to_string(next_) + "\n";
    str += std::string(kSerializeSolverEpsilonSectionHeader) + "\n" + FormatDouble(epsilon_, -1) + "\n";
    str += std::string(kSerializeSolverDefaultPolicySectionHeader) + "\nUniformPolicy:\n";
    str += std::string(kSerializeSolverValuesTableSectionHeader) + "\n";
    return str + SerializeValuesTable(InfoStateValuesTable(), double_precision, delimiter);
  }
  void RestoreCounter(uint64_t seed, int64_t next) { seed_ = seed; next_ = next; }
  void RestoreGenerator(const std::string& state) {  // outcome_sampling_mccfr.cc:281-283
    std::istringstream in(state);
    in >> rng_;
    if (in.fail()) SpielFatalError("[SolverRNG] does not hold a std::mt19937 state");
  }
  int64_t EpisodesRun() const { return next_; }
  double Epsilon() const { return epsilon_; }

 private:
  uint64_t seed_;
  int64_t next_ = 0;
  std::mt19937 rng_;
  double epsilon_;
  std::string game_string_;
};

namespace internal {
// The value line that follows `header` inside [SolverSpecificState].
inline std::string SpecificLine(const PartialCheckpoint& c, const char* header) {
  for (size_t i = 0; i + 1 < c.specific.size(); ++i)
    if (c.specific[i] == header) return c.specific[i + 1];
  SpielFatalError(std::string("solver checkpoint without a ") + header + " section");
}
inline void ParseCounter(const std::string& line, uint64_t* seed, int64_t* next) {
  char tag[16] = {0};
  unsigned long long s = 0;
  long long n = 0;
  if (std::sscanf(line.c_str(), "%15s %llu %lld", tag, &s, &n) != 3 || std::string(tag) != "counter")
    SpielFatalError("[SolverRNG] is not a counter-RNG position (a reference mt19937 dump cannot be continued here)");
  *seed = s;
  *next = n;
}
}  // namespace internal

inline std::unique_ptr<ExternalSamplingMCCFRSolver> DeserializeExternalSamplingMCCFRSolver(
    const std::string& serialized, const std::string& delimiter = "<~>") {  // external_sampling_mccfr.cc:233-288
  const PartialCheckpoint c = PartiallyDeserializeSolver(serialized);
  if (c.solver_type != "ExternalSamplingMCCFRSolver")
    SpielFatalError("checkpoint holds a " + c.solver_type + ", not an ExternalSamplingMCCFRSolver");
  // [SolverRNG]: the reference's mt19937 dump (a checkpoint written by the reference loads and continues draw
  // for draw), optionally followed by this engine's "counter <seed> <next>" line (older checkpoints of this
  // engine hold that line alone).
  uint64_t seed = 0;
  int64_t next = 0;
  std::string generator;
  {
    size_t i = 0;
    while (i < c.specific.size() && c.specific[i] != kSerializeSolverRNGSectionHeader) ++i;
    if (i == c.specific.size()) SpielFatalError("solver checkpoint without a [SolverRNG] section");
    for (++i; i < c.specific.size() && c.specific[i][0] != '['; ++i) {
      if (c.specific[i].rfind("counter ", 0) == 0) internal::ParseCounter(c.specific[i], &seed, &next);
      else generator += c.specific[i] + " ";
    }
  }
  const std::string avg = internal::SpecificLine(c, kSerializeSolverAverageTypeSectionHeader);
  if (avg != "SimpleAverageType" && avg != "FullAverageType") SpielFatalError("unknown average type " + avg);
  std::shared_ptr<const Game> game = LoadGame(c.game);
  auto solver = std::make_unique<ExternalSamplingMCCFRSolver>(
      *game, static_cast<int>(seed), avg == "FullAverageType" ? AverageType::kFull : AverageType::kSimple);
  solver->RestoreCounter(seed, next);
  if (!generator.empty()) solver->RestoreGenerator(generator);
  solver->LoadInfoStateValuesTable(DeserializeValuesTable(c.table, delimiter), /*allow_missing=*/true);
  return solver;
}
inline std::unique_ptr<OutcomeSamplingMCCFRSolver> DeserializeOutcomeSamplingMCCFRSolver(
    const std::string& serialized, const std::string& delimiter = "<~>") {  // outcome_sampling_mccfr.cc:243-300
  const PartialCheckpoint c = PartiallyDeserializeSolver(serialized);
  if (c.solver_type != "OutcomeSamplingMCCFRSolver")
    SpielFatalError("checkpoint holds a " + c.solver_type + ", not an OutcomeSamplingMCCFRSolver");
  // [SolverRNG]: the reference's mt19937 dump (a checkpoint written by the reference loads here and vice versa),
  // optionally followed by this engine's "counter <seed> <next>" line (older checkpoints hold that line alone).
  uint64_t seed = 0;
  int64_t next = 0;
  std::string generator;
  {
    size_t i = 0;
    while (i < c.specific.size() && c.specific[i] != kSerializeSolverRNGSectionHeader) ++i;
    if (i == c.specific.size()) SpielFatalError("solver checkpoint without a [SolverRNG] section");
    for (++i; i < c.specific.size() && c.specific[i][0] != '['; ++i) {
      if (c.specific[i].rfind("counter ", 0) == 0) internal::ParseCounter(c.specific[i], &seed, &next);
      else generator += c.specific[i] + " ";
    }
  }
  const double epsilon = std::strtod(internal::SpecificLine(c, kSerializeSolverEpsilonSectionHeader).c_str(), nullptr);
  std::shared_ptr<const Game> game = LoadGame(c.game);
  auto solver = std::make_unique<OutcomeSamplingMCCFRSolver>(*game, epsilon, static_cast<int>(seed));
  solver->RestoreCounter(seed, next);
  if (!generator.empty()) solver->RestoreGenerator(generator);
  solver->LoadInfoStateValuesTable(DeserializeValuesTable(c.table, delimiter), /*allow_missing=*/true);
  return solver;
}

}  // namespace algorithms
}  // namespace hip
}  // namespace open_spiel

#endif  // OSG_HOST_OSG_SPIEL_H_
