// pybind11 module `open_spiel_amd.pyspiel_hip`: the pyspiel surface of the hot path
// (open_spiel/python/pybind11/pyspiel.cc:356-533,720-731 State / Game / load_game;
//  bots.cc:106-149 Evaluator / RandomRolloutEvaluator / SearchNode / MCTSBot;
//  policy.cc:224-333 CFRSolver / CFRPlusSolver / ExternalSamplingMCCFRSolver)
// over the C++ host mirror (osg_spiel.h), i.e. over the C-ABI of libosg_hip.so.
// Same method names and argument meaning as pyspiel, so scripts written against
// `import pyspiel` run with `from open_spiel_amd import pyspiel_hip as pyspiel` for the five
// hot-path games.  Plus the batch classes the device actually wants (BatchedState, step_batch).
#include <pybind11/functional.h>
#include <pybind11/numpy.h>
#include <sstream>
#include <optional>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "osg_spiel.h"

namespace py = pybind11;
using namespace open_spiel::hip;
using namespace open_spiel::hip::algorithms;

namespace {

template <class T>
py::array_t<T> as_array(const std::vector<T>& v, std::vector<py::ssize_t> shape) {
  py::array_t<T> a(shape);
  std::memcpy(a.mutable_data(), v.data(), v.size() * sizeof(T));
  return a;
}

}  // namespace


// Game::GetParameters (spiel.h:1013): the given parameters plus every default, typed like pyspiel does.
py::dict GameParametersWithDefaults(const std::string& game_string) {
  const size_t open = game_string.find('(');
  const std::string name = game_string.substr(0, open);
  std::vector<std::pair<std::string, std::string>> kv;  // defaults first, then the given values
  if (name == "connect_four") kv = {{"columns", "7"}, {"egocentric_obs_tensor", "False"}, {"rows", "6"}, {"x_in_row", "4"}};
  else if (name == "hex") kv = {{"board_size", "11"}, {"num_cols", ""}, {"num_rows", ""}, {"plain_obs_tensor", "False"},
                                {"string_rep", "standard"}, {"swap", "False"}};
  else if (name == "kuhn_poker") kv = {{"players", "2"}};
  else if (name == "leduc_poker") kv = {{"action_mapping", "False"}, {"players", "2"}, {"starting_player", "0"},
                                        {"suit_isomorphism", "False"}};
  if (open != std::string::npos) {
    const std::string body = game_string.substr(open + 1, game_string.rfind(')') - open - 1);
    size_t pos = 0;
    while (pos < body.size()) {
      size_t comma = body.find(',', pos);
      if (comma == std::string::npos) comma = body.size();
      const std::string item = body.substr(pos, comma - pos);
      pos = comma + 1;
      const size_t eq = item.find('=');
      if (eq == std::string::npos) continue;
      const std::string k = item.substr(0, eq), v = item.substr(eq + 1);
      bool found = false;
      for (auto& e : kv)
        if (e.first == k) { e.second = v; found = true; }
      if (!found) kv.emplace_back(k, v);
    }
  }
  if (name == "hex") {  // num_cols / num_rows default to board_size (hex.cc:60-63)
    std::string bs;
    for (auto& e : kv) if (e.first == "board_size") bs = e.second;
    for (auto& e : kv) if ((e.first == "num_cols" || e.first == "num_rows") && e.second.empty()) e.second = bs;
  }
  py::dict out;
  for (const auto& e : kv) {
    const std::string& v = e.second;
    if (v == "True" || v == "true") out[py::str(e.first)] = py::bool_(true);
    else if (v == "False" || v == "false") out[py::str(e.first)] = py::bool_(false);
    else if (!v.empty() && v.find_first_not_of("-0123456789") == std::string::npos) out[py::str(e.first)] = py::int_(std::stoll(v));
    else out[py::str(e.first)] = py::str(v);
  }
  return out;
}

// "name(k=v,...)" from a short name and a Python dict of parameters: bools as True / False, the rest through str()
std::string GameStringFromDict(const std::string& name, const py::dict& params) {
  std::vector<std::string> items;
  for (auto item : params) {
    const std::string key = py::cast<std::string>(item.first);
    std::string value;
    if (py::isinstance<py::bool_>(item.second)) value = py::cast<bool>(item.second) ? "True" : "False";
    else value = py::cast<std::string>(py::str(item.second));
    items.push_back(key + "=" + value);
  }
  std::sort(items.begin(), items.end());
  if (items.empty()) return name;
  std::string out = name + "(";
  for (size_t i = 0; i < items.size(); ++i) out += (i ? "," : "") + items[i];
  return out + ")";
}
// only the parameters a game string gives, typed like get_parameters() types them
py::dict GameParametersGiven(const std::string& game_string) {
  py::dict out;
  const size_t open = game_string.find('(');
  if (open == std::string::npos) return out;
  const std::string body = game_string.substr(open + 1, game_string.rfind(')') - open - 1);
  size_t pos = 0;
  while (pos < body.size()) {
    size_t comma = body.find(',', pos);
    if (comma == std::string::npos) comma = body.size();
    const std::string item = body.substr(pos, comma - pos);
    pos = comma + 1;
    const size_t eq = item.find('=');
    if (eq == std::string::npos) continue;
    const std::string k = item.substr(0, eq), v = item.substr(eq + 1);
    if (v == "True" || v == "true") out[py::str(k)] = py::bool_(true);
    else if (v == "False" || v == "false") out[py::str(k)] = py::bool_(false);
    else if (!v.empty() && v.find_first_not_of("-0123456789") == std::string::npos) out[py::str(k)] = py::int_(std::stoll(v));
    else out[py::str(k)] = py::str(v);
  }
  return out;
}

struct GameInfoRecord {  // spiel.h:178-215 GameInfo: what a Python game module describes itself with
  int num_distinct_actions, max_chance_outcomes, num_players;
  double min_utility, max_utility;
  std::optional<double> utility_sum;
  int max_game_length;
  GameInfoRecord(int a, int c, int p, double lo, double hi, std::optional<double> sum, int len)
      : num_distinct_actions(a), max_chance_outcomes(c), num_players(p), min_utility(lo), max_utility(hi), utility_sum(sum),
        max_game_length(len) {}
};
enum class TttCellState { kEmpty, kNought, kCross };          // tic_tac_toe.h:38-42
enum class LeducActionType { kFold = 0, kCall = 1, kRaise = 2 };  // leduc_poker.h:64

static std::string ShortName(const State& st) {
  const std::string text = st.GetGame()->ToString();
  return text.substr(0, text.find('('));
}
// TicTacToeState::Board (tic_tac_toe.h:97): row-major cells, read off the state's string (tic_tac_toe.cc:163-175)
static std::vector<TttCellState> TttBoard(const State& st) {
  if (ShortName(st) != "tic_tac_toe") SpielFatalError("board(): a tic_tac_toe accessor");
  std::vector<TttCellState> out;
  for (char c : st.ToString()) {
    if (c == '.') out.push_back(TttCellState::kEmpty);
    else if (c == 'o') out.push_back(TttCellState::kNought);
    else if (c == 'x') out.push_back(TttCellState::kCross);
  }
  if (out.size() != 9) SpielFatalError("tic_tac_toe board: unexpected state string");
  return out;
}
// The fields LeducState exposes to Python (leduc_poker.h:120-135), parsed from LeducState::ToString
// (leduc_poker.cc:463-496): "Round: r\nPlayer: p\nPot: n\nMoney (p1 p2 ...): a b\nCards (public p1 p2 ...): c a b \n
// Round 1 sequence: Call, Raise\nRound 2 sequence: ...\n"
struct LeducFields {
  int round = 0, pot = 0, public_card = -10000;
  std::vector<int> money, private_cards, round1, round2;
};
static LeducFields LeducView(const State& st) {
  if (ShortName(st) != "leduc_poker") SpielFatalError("a leduc_poker accessor");
  LeducFields f;
  std::istringstream in(st.ToString());
  std::string line;
  auto ints_after_colon = [](const std::string& l) {
    std::vector<int> v;
    std::istringstream is(l.substr(l.find(':') + 1));
    int x;
    while (is >> x) v.push_back(x);
    return v;
  };
  auto actions_after_colon = [](const std::string& l) {
    std::vector<int> v;
    std::string rest = l.substr(l.find(':') + 1), tok;
    std::istringstream is(rest);
    while (std::getline(is, tok, ',')) {
      if (tok.find("Fold") != std::string::npos) v.push_back(0);
      else if (tok.find("Call") != std::string::npos) v.push_back(1);
      else if (tok.find("Raise") != std::string::npos) v.push_back(2);
    }
    return v;
  };
  while (std::getline(in, line)) {
    if (line.rfind("Round:", 0) == 0) f.round = ints_after_colon(line).at(0);
    else if (line.rfind("Pot:", 0) == 0) f.pot = ints_after_colon(line).at(0);
    else if (line.rfind("Money", 0) == 0) f.money = ints_after_colon(line);
    else if (line.rfind("Cards", 0) == 0) {
      std::vector<int> c = ints_after_colon(line);
      if (!c.empty()) { f.public_card = c[0]; f.private_cards.assign(c.begin() + 1, c.end()); }
    } else if (line.rfind("Round 1 sequence", 0) == 0) f.round1 = actions_after_colon(line);
    else if (line.rfind("Round 2 sequence", 0) == 0) f.round2 = actions_after_colon(line);
  }
  return f;
}

// Python subclasses of Policy: the reference's python/policy.py interface, action_probabilities(state, player_id)
class PyPolicy : public Policy {
 public:
  using Policy::Policy;
  ActionsAndProbs GetStatePolicy(const State& state, Player player) const override {
    py::gil_scoped_acquire gil;
    py::function f = py::get_override(static_cast<const Policy*>(this), "action_probabilities");
    if (!f) SpielFatalError("a Python Policy must define action_probabilities(state, player_id)");
    py::dict d = f(py::cast(state, py::return_value_policy::reference), player);
    ActionsAndProbs ap;
    for (auto kv : d) ap.push_back({kv.first.cast<Action>(), kv.second.cast<double>()});
    std::sort(ap.begin(), ap.end());
    return ap;
  }
  ActionsAndProbs GetStatePolicy(const State& state) const override { return GetStatePolicy(state, state.CurrentPlayer()); }
};

// Python subclasses of Evaluator (mcts.h:83-92)
class PyEvaluator : public Evaluator {
 public:
  using Evaluator::Evaluator;
  std::vector<double> Evaluate(const State& state) override {
    PYBIND11_OVERRIDE_PURE_NAME(std::vector<double>, Evaluator, "evaluate", Evaluate, state);
  }
  ActionsAndProbs Prior(const State& state) override {
    PYBIND11_OVERRIDE_PURE_NAME(ActionsAndProbs, Evaluator, "prior", Prior, state);
  }
};

// Python subclasses of Bot (python/pybind11/python_bots.h: PyBot; e.g. open_spiel/python/bots/uniform_random.py)
class PyBot : public Bot {
 public:
  using Bot::Bot;
  Action Step(const State& state) override { PYBIND11_OVERRIDE_PURE_NAME(Action, Bot, "step", Step, state); }
  void Restart() override { PYBIND11_OVERRIDE_NAME(void, Bot, "restart", Restart); }
  void RestartAt(const State& state) override { PYBIND11_OVERRIDE_NAME(void, Bot, "restart_at", RestartAt, state); }
  void InformAction(const State& state, Player player_id, Action action) override {
    PYBIND11_OVERRIDE_NAME(void, Bot, "inform_action", InformAction, state, player_id, action);
  }
  void InformActions(const State& state, const std::vector<Action>& actions) override {
    PYBIND11_OVERRIDE_NAME(void, Bot, "inform_actions", InformActions, state, actions);
  }
  bool ProvidesForceAction() override { PYBIND11_OVERRIDE_NAME(bool, Bot, "provides_force_action", ProvidesForceAction); }
  void ForceAction(const State& state, Action action) override {
    PYBIND11_OVERRIDE_NAME(void, Bot, "force_action", ForceAction, state, action);
  }
  bool ProvidesPolicy() override { PYBIND11_OVERRIDE_NAME(bool, Bot, "provides_policy", ProvidesPolicy); }
  ActionsAndProbs GetPolicy(const State& state) override { PYBIND11_OVERRIDE_NAME(ActionsAndProbs, Bot, "get_policy", GetPolicy, state); }
  std::pair<ActionsAndProbs, Action> StepWithPolicy(const State& state) override {
    using StepRet = std::pair<ActionsAndProbs, Action>;
    PYBIND11_OVERRIDE_NAME(StepRet, Bot, "step_with_policy", StepWithPolicy, state);
  }
};

// python/pybind11/pybind11.h:198-220: a struct type with constructors from nothing, a JSON string and a dict
template <typename StructType, typename BaseType>
auto bind_spiel_struct(py::module_& m, const char* name) {
  return py::class_<StructType, BaseType>(m, name)
      .def(py::init<>())
      .def(py::init<std::string>())
      .def(py::init([](py::dict d) { return StructType(py::module_::import("json").attr("dumps")(d).cast<std::string>()); }));
}

PYBIND11_MODULE(pyspiel_hip, m) {
  m.doc() = "pyspiel-compatible surface of the MI355X game-step and search engine (libosg_hip.so)";
  py::register_exception<SpielException>(m, "SpielError", PyExc_RuntimeError);  // pyspiel.cc:831-837

  m.attr("INVALID_ACTION") = py::int_(kInvalidAction);
  {  // pyspiel.cc:139-158: PlayerId as a Python IntEnum (spiel_globals.h:44-60), so that state.current_player() == PlayerId.CHANCE
    py::dict members;
    members["DEFAULT_PLAYER_ID"] = py::int_(kDefaultPlayerId);
    members["INVALID"] = py::int_(kInvalidPlayer);
    members["TERMINAL"] = py::int_(kTerminalPlayerId);
    members["CHANCE"] = py::int_(kChancePlayerId);
    members["MEAN_FIELD"] = py::int_(kMeanFieldPlayerId);
    members["SIMULTANEOUS"] = py::int_(kSimultaneousPlayerId);
    m.attr("PlayerId") = py::module_::import("enum").attr("IntEnum")("PlayerId", members, py::arg("module") = py::str(m.attr("__name__")));
  }
  py::enum_<TensorLayout>(m, "TensorLayout")  // pyspiel.cc:330-333
      .value("HWC", TensorLayout::kHWC)
      .value("CHW", TensorLayout::kCHW);
  py::class_<State::PlayerAction>(m, "PlayerAction")  // pyspiel.cc:343-345
      .def_readonly("player", &State::PlayerAction::player)
      .def_readonly("action", &State::PlayerAction::action);
  py::class_<Game, std::shared_ptr<Game>>(m, "Game")
      .def("num_distinct_actions", &Game::NumDistinctActions)
      .def("max_chance_outcomes", &Game::MaxChanceOutcomes)
      .def("num_players", &Game::NumPlayers)
      .def("min_utility", &Game::MinUtility)
      .def("max_utility", &Game::MaxUtility)
      .def("max_game_length", &Game::MaxGameLength)
      .def("max_chance_nodes_in_history", &Game::MaxChanceNodesInHistory)
      .def("max_move_number", &Game::MaxMoveNumber)
      .def("max_history_length", &Game::MaxHistoryLength)
      .def("policy_tensor_shape", &Game::PolicyTensorShape)
      .def("observation_tensor_layout", &Game::ObservationTensorLayout)
      .def("information_state_tensor_layout", &Game::InformationStateTensorLayout)
      .def("observation_tensor_shape", &Game::ObservationTensorShape)
      .def("observation_tensor_size", &Game::ObservationTensorSize)
      .def("information_state_tensor_shape", &Game::InformationStateTensorShape)
      .def("information_state_tensor_size", &Game::InformationStateTensorSize)
      .def("new_initial_state", [](const Game& g) { return g.NewInitialState(); })
      // pyspiel.cc:484-494: from a JSON string, a dict, a StateStruct
      .def("new_initial_state", [](const Game& g, const std::string& json) { return g.NewInitialState(Json::parse(json)); })
      .def("new_initial_state",
           [](const Game& g, const py::dict& d) {
             return g.NewInitialState(Json::parse(py::module_::import("json").attr("dumps")(d).cast<std::string>()));
           })
      .def("new_initial_state", [](const Game& g, const StateStruct& s) { return g.NewInitialState(s); })
      .def("serialize", &Game::Serialize)
      .def("deserialize_state", &Game::DeserializeState, py::arg("serialized"))
      .def(py::pickle([](const Game& g) { return g.Serialize(); },  // pyspiel.cc:535-543
                      [](const std::string& t) { return std::const_pointer_cast<Game>(LoadGame(t)); }))
      .def("utility_sum", [](const Game&) { return 0.0; })  // all five games are zero-sum (spiel.h:1006-1011)
      .def("get_parameters", [](const Game& g) { return GameParametersWithDefaults(g.ToString()); })
      .def("get_type", &Game::GetType)
      .def("new_initial_states", [](const Game& g, int64_t n) { return g.NewInitialStates(n); }, py::arg("n"))
      // pyspiel.cc:509-533: make_observer(iig_obs_type=None, params={}) -> Observer or None
      .def("make_observer",
           [](const Game& g, std::optional<IIGObservationType> t, const py::dict& params) -> std::shared_ptr<Observer> {
             if (params.size()) SpielFatalError("Observation parameters not supported");
             return MakeObserver(g, t ? &*t : nullptr);
           },
           py::arg("imperfect_information_observation_type") = py::none(), py::arg("params") = py::dict())
      .def("__str__", &Game::ToString)
      .def("__repr__", &Game::ToString);
  py::class_<GameType> game_type(m, "GameType");  // pyspiel.cc:246-303: the fields scripts look at, with the nested enums
  py::enum_<GameType::Dynamics>(game_type, "Dynamics")
      .value("SEQUENTIAL", GameType::Dynamics::kSequential)
      .value("MEAN_FIELD", GameType::Dynamics::kMeanField)
      .value("SIMULTANEOUS", GameType::Dynamics::kSimultaneous);
  py::enum_<GameType::ChanceMode>(game_type, "ChanceMode")
      .value("DETERMINISTIC", GameType::ChanceMode::kDeterministic)
      .value("EXPLICIT_STOCHASTIC", GameType::ChanceMode::kExplicitStochastic)
      .value("SAMPLED_STOCHASTIC", GameType::ChanceMode::kSampledStochastic);
  py::enum_<GameType::Information>(game_type, "Information")
      .value("ONE_SHOT", GameType::Information::kOneShot)
      .value("PERFECT_INFORMATION", GameType::Information::kPerfectInformation)
      .value("IMPERFECT_INFORMATION", GameType::Information::kImperfectInformation);
  py::enum_<GameType::Utility>(game_type, "Utility")
      .value("ZERO_SUM", GameType::Utility::kZeroSum)
      .value("CONSTANT_SUM", GameType::Utility::kConstantSum)
      .value("GENERAL_SUM", GameType::Utility::kGeneralSum)
      .value("IDENTICAL", GameType::Utility::kIdentical);
  py::enum_<GameType::RewardModel>(game_type, "RewardModel")
      .value("REWARDS", GameType::RewardModel::kRewards)
      .value("TERMINAL", GameType::RewardModel::kTerminal);
  // the constructor Python game modules call at import (open_spiel/python/games/*.py build a GameType and a GameInfo and
  // hand them to register_game): constructible here so that those modules import; the engine does not serve Python games
  game_type.def(py::init([](const std::string& short_name, const std::string& long_name, GameType::Dynamics dynamics,
                            GameType::ChanceMode chance_mode, GameType::Information information, GameType::Utility utility,
                            GameType::RewardModel reward_model, int max_num_players, int min_num_players,
                            bool provides_information_state_string, bool provides_information_state_tensor,
                            bool provides_observation_string, bool provides_observation_tensor, const py::dict&,
                            bool default_loadable, bool provides_factored_observation_string, bool) {
                  GameType t;
                  t.short_name = short_name; t.long_name = long_name; t.dynamics = dynamics; t.chance_mode = chance_mode;
                  t.information = information; t.utility = utility; t.reward_model = reward_model;
                  t.max_num_players = max_num_players; t.min_num_players = min_num_players;
                  t.provides_information_state_string = provides_information_state_string;
                  t.provides_information_state_tensor = provides_information_state_tensor;
                  t.provides_observation_string = provides_observation_string;
                  t.provides_observation_tensor = provides_observation_tensor;
                  t.default_loadable = default_loadable;
                  t.provides_factored_observation_string = provides_factored_observation_string;
                  return t;
                }),
                py::arg("short_name"), py::arg("long_name"), py::arg("dynamics"), py::arg("chance_mode"), py::arg("information"),
                py::arg("utility"), py::arg("reward_model"), py::arg("max_num_players"), py::arg("min_num_players"),
                py::arg("provides_information_state_string"), py::arg("provides_information_state_tensor"),
                py::arg("provides_observation_string"), py::arg("provides_observation_tensor"),
                py::arg("parameter_specification") = py::dict(), py::arg("default_loadable") = true,
                py::arg("provides_factored_observation_string") = false, py::arg("action_structs_only") = false);
  py::class_<GameInfoRecord>(m, "GameInfo")  // pyspiel.cc:305-320
      .def(py::init<int, int, int, double, double, std::optional<double>, int>(), py::arg("num_distinct_actions"),
           py::arg("max_chance_outcomes"), py::arg("num_players"), py::arg("min_utility"), py::arg("max_utility"),
           py::arg("utility_sum") = std::nullopt, py::arg("max_game_length"))
      .def_readonly("num_distinct_actions", &GameInfoRecord::num_distinct_actions)
      .def_readonly("max_chance_outcomes", &GameInfoRecord::max_chance_outcomes)
      .def_readonly("num_players", &GameInfoRecord::num_players)
      .def_readonly("min_utility", &GameInfoRecord::min_utility)
      .def_readonly("max_utility", &GameInfoRecord::max_utility)
      .def_readonly("utility_sum", &GameInfoRecord::utility_sum)
      .def_readonly("max_game_length", &GameInfoRecord::max_game_length);
  m.def("register_game",  // pyspiel.cc:798: accepted and remembered (registered_python_games()); never loadable here
        [](const GameType& type, py::object creator) {
          static py::dict* registry = new py::dict();
          (*registry)[py::str(type.short_name)] = creator;
          py::module_::import("sys").attr("modules")[py::str("open_spiel_amd.pyspiel_hip")].attr("_python_games") = *registry;
        },
        py::arg("game_type"), py::arg("creator"));
  game_type.def_readonly("short_name", &GameType::short_name)
      .def_readonly("long_name", &GameType::long_name)
      .def_readonly("dynamics", &GameType::dynamics)
      .def_readonly("chance_mode", &GameType::chance_mode)
      .def_readonly("information", &GameType::information)
      .def_readonly("utility", &GameType::utility)
      .def_readonly("reward_model", &GameType::reward_model)
      .def_readonly("max_num_players", &GameType::max_num_players)
      .def_readonly("min_num_players", &GameType::min_num_players)
      .def_readonly("provides_information_state_string", &GameType::provides_information_state_string)
      .def_readonly("provides_information_state_tensor", &GameType::provides_information_state_tensor)
      .def_readonly("provides_observation_string", &GameType::provides_observation_string)
      .def_readonly("provides_observation_tensor", &GameType::provides_observation_tensor)
      .def_readonly("provides_factored_observation_string", &GameType::provides_factored_observation_string)
      .def_readonly("default_loadable", &GameType::default_loadable)
      .def_readonly("is_concrete", &GameType::is_concrete)
      .def("provides_information_state", &GameType::provides_information_state)
      .def("provides_observation", &GameType::provides_observation)
      .def("__repr__", [](const GameType& t) { return "<GameType '" + t.short_name + "'>"; });
  m.def("load_game", [](const std::string& s) { return std::make_shared<Game>(s); });  // pyspiel.cc:720-731
  m.def("load_game", [](const GameParametersStruct& p) { return std::const_pointer_cast<Game>(LoadGame(p)); });  // pyspiel.cc:732-733
  m.def("load_game_from_json", [](const std::string& json) { return std::const_pointer_cast<Game>(LoadGameFromJson(json)); });
  // load_game(short_name, {"players": 3, "swap": True}) (pyspiel.cc:732-741): the parameters as a Python dict
  m.def("load_game",
        [](const std::string& name, const py::dict& params) { return std::make_shared<Game>(GameStringFromDict(name, params)); },
        py::arg("short_name"), py::arg("params"));
  m.def("registered_names", [] { return std::vector<std::string>{"connect_four", "hex", "kuhn_poker", "leduc_poker", "tic_tac_toe"}; });
  m.def("registered_games", [] {  // pyspiel.cc:774: the GameType of every game that can be loaded
    std::vector<GameType> out;
    for (const char* name : {"connect_four", "hex", "kuhn_poker", "leduc_poker", "tic_tac_toe"}) out.push_back(Game(name).GetType());
    return out;
  });
  m.def("registered_concrete_names",
        [] { return std::vector<std::string>{"connect_four", "hex", "kuhn_poker", "leduc_poker", "tic_tac_toe"}; });
  m.def("game_parameters_from_string",  // pyspiel.cc:168: "kuhn_poker(players=3)" -> {"name": "kuhn_poker", "players": 3}
        [](const std::string& game_string) {
          py::dict out = GameParametersGiven(game_string);
          out[py::str("name")] = py::str(game_string.substr(0, game_string.find('(')));
          return out;
        });
  m.def("game_parameters_to_string", [](const py::dict& params) {  // pyspiel.cc:171
    if (!params.contains("name")) SpielFatalError("game_parameters_to_string: the dictionary has no \"name\"");
    py::dict rest;
    for (auto item : params)
      if (py::cast<std::string>(item.first) != "name") rest[item.first] = item.second;
    return GameStringFromDict(py::cast<std::string>(params["name"]), rest);
  });
  m.def("sample_action",  // pyspiel.cc:811-815, spiel_utils SampleAction(outcomes, z): cumulative intervals, z in [0, 1)
        [](const ActionsAndProbs& outcomes, double z) {
          if (!(z >= 0.0 && z < 1.0)) SpielFatalError("sample_action: z must be in [0, 1)");
          double acc = 0;
          for (const auto& ap : outcomes) {
            if (z >= acc && z < acc + ap.second) return ap;
            acc += ap.second;
          }
          SpielFatalError("sample_action: the probabilities do not cover z");
        },
        py::arg("actions_and_probs"), py::arg("z"));
  py::enum_<StateType>(m, "StateType")  // pyspiel.cc:317-322 (spiel_globals.h:84-92)
      .value("TERMINAL", StateType::kTerminal)
      .value("CHANCE", StateType::kChance)
      .value("DECISION", StateType::kDecision)
      .value("MEAN_FIELD", StateType::kMeanField);

  // ---- the struct API (pyspiel.cc:322-345; python/pybind11/pybind11.h:198-220 bind_spiel_struct): to_json() / to_dict(),
  // constructors from nothing, a JSON string or a dict ----
  py::class_<SpielStruct>(m, "SpielStruct")
      .def("to_json", &SpielStruct::ToJson)
      .def("to_dict", [](const SpielStruct& self) { return py::module_::import("json").attr("loads")(self.ToJson()); })
      .def("__str__", &SpielStruct::ToJson);
  py::class_<StateStruct, SpielStruct>(m, "StateStruct");
  py::class_<ObservationStruct, SpielStruct>(m, "ObservationStruct");
  py::class_<ActionStruct, SpielStruct>(m, "ActionStruct");
  py::class_<GameParametersStruct, SpielStruct>(m, "GameParametersStruct").def_readwrite("game_name", &GameParametersStruct::game_name);
  py::class_<Status>(m, "Status")  // utils/status.h
      .def("ok", &Status::ok)
      .def("message", &Status::message)
      .def("to_string", &Status::ToString)
      .def("__bool__", &Status::ok)
      .def("__str__", &Status::ToString);
  py::class_<State>(m, "State")
      .def("current_player", &State::CurrentPlayer)
      .def("is_terminal", &State::IsTerminal)
      .def("is_chance_node", &State::IsChanceNode)
      .def("legal_actions", py::overload_cast<>(&State::LegalActions, py::const_))
      .def("legal_actions", py::overload_cast<Player>(&State::LegalActions, py::const_), py::arg("player"))
      .def("legal_actions_mask", py::overload_cast<>(&State::LegalActionsMask, py::const_))
      .def("legal_actions_mask", py::overload_cast<Player>(&State::LegalActionsMask, py::const_), py::arg("player"))
      .def("apply_action_with_legality_check", &State::ApplyActionWithLegalityCheck, py::arg("action"))
      .def("undo_action", &State::UndoAction, py::arg("player"), py::arg("action"))
      .def("is_simultaneous_node", &State::IsSimultaneousNode)
      .def("is_player_node", &State::IsPlayerNode)
      .def("is_initial_state", &State::IsInitialState)
      .def("resample_from_infostate",
           [](const State& s, int player, std::function<double()> sampler) { return s.ResampleFromInfostate(player, sampler); },
           py::arg("player_id"), py::arg("probability_sampler"))
      .def("apply_action", &State::ApplyAction, py::arg("action"))
      .def("returns", &State::Returns)
      .def("rewards", &State::Rewards)
      .def("player_return", &State::PlayerReturn, py::arg("player"))
      .def("chance_outcomes", &State::ChanceOutcomes)
      .def("observation_tensor", [](const State& s, Player p) { return s.ObservationTensor(p); }, py::arg("player"))
      .def("observation_tensor", [](const State& s) { return s.ObservationTensor(std::max(s.CurrentPlayer(), 0)); })
      .def("information_state_tensor", [](const State& s, Player p) { return s.InformationStateTensor(p); },
           py::arg("player"))
      .def("information_state_tensor",
           [](const State& s) { return s.InformationStateTensor(std::max(s.CurrentPlayer(), 0)); })
      .def("information_state_string", py::overload_cast<Player>(&State::InformationStateString, py::const_),
           py::arg("player"))
      .def("information_state_string", py::overload_cast<>(&State::InformationStateString, py::const_))
      .def("observation_string", py::overload_cast<Player>(&State::ObservationString, py::const_), py::arg("player"))
      .def("observation_string", py::overload_cast<>(&State::ObservationString, py::const_))
      // pyspiel.cc:364-411, 445-450, 473-477: the struct API (tic_tac_toe and connect_four)
      .def("to_struct", &State::ToStruct)
      .def("to_json", &State::ToJson)
      .def("to_dict", [](const State& st) { return py::module_::import("json").attr("loads")(st.ToJson()); })
      .def("to_observation_struct", py::overload_cast<Player>(&State::ToObservationStruct, py::const_), py::arg("player"))
      .def("to_observation_struct", py::overload_cast<>(&State::ToObservationStruct, py::const_))
      .def("action_to_struct", py::overload_cast<Player, Action>(&State::ActionToStruct, py::const_))
      .def("action_to_struct", py::overload_cast<Action>(&State::ActionToStruct, py::const_))
      .def("struct_to_actions", &State::StructToActions)
      .def("actions_to_struct", py::overload_cast<Player, const std::vector<Action>&>(&State::ActionsToStruct, py::const_))
      .def("actions_to_struct", py::overload_cast<const std::vector<Action>&>(&State::ActionsToStruct, py::const_))
      .def("validate_action_struct", &State::ValidateActionStruct)
      .def("apply_action_struct", &State::ApplyActionStruct)
      .def("__str__", &State::ToString)
      .def("to_string", &State::ToString)
      .def("action_to_string", py::overload_cast<Player, Action>(&State::ActionToString, py::const_),
           py::arg("player"), py::arg("action"))
      .def("action_to_string", py::overload_cast<Action>(&State::ActionToString, py::const_), py::arg("action"))
      .def("history_str", &State::HistoryString)
      .def("clone", &State::Clone)
      .def("child", &State::Child, py::arg("action"))
      .def("history", &State::History)
      .def("full_history", &State::FullHistory)                     // pyspiel.cc:426
      .def("get_type", &State::GetType)                             // spiel.h:808
      .def("string_to_action", py::overload_cast<Player, const std::string&>(&State::StringToAction, py::const_), py::arg("player"),
           py::arg("string"))
      .def("string_to_action", py::overload_cast<const std::string&>(&State::StringToAction, py::const_), py::arg("string"))
      .def("is_initial_non_chance_state", &State::IsInitialNonChanceState)
      .def("is_mean_field_node", &State::IsMeanFieldNode)
      .def("player_reward", &State::PlayerReward, py::arg("player"))
      .def("apply_actions", &State::ApplyActions, py::arg("actions"))  // sequential games: fatal, as in the reference
      .def("apply_actions_with_legality_checks", &State::ApplyActionsWithLegalityChecks, py::arg("actions"))
      .def("distribution_support", &State::DistributionSupport)
      .def("update_distribution", &State::UpdateDistribution, py::arg("distribution"))
      .def("serialize", &State::Serialize)
      .def("starting_state", &State::StartingState)          // pyspiel.cc:459-460
      .def("starting_state_str", &State::StartingStateStr)
      .def(py::pickle(  // pyspiel.cc:455-474: a state pickles as its game-and-state text
          [](const State& s) { return SerializeGameAndState(*s.GetGame(), s); },
          [](const std::string& t) { return std::move(DeserializeGameAndState(t).second); }))
      // tic_tac_toe.TicTacToeState (games_tic_tac_toe.cc:80-86)
      .def("board", [](const State& st) { return TttBoard(st); })
      .def("board_at", [](const State& st, int row, int col) {
             if (row < 0 || row > 2 || col < 0 || col > 2) SpielFatalError("board_at: out of range");
             return TttBoard(st)[row * 3 + col];
           }, py::arg("row"), py::arg("col"))
      // leduc_poker.LeducState (games_leduc_poker.cc:40-50)
      .def("get_private_cards", [](const State& st) { return LeducView(st).private_cards; })
      .def("private_card", [](const State& st, Player p) { return LeducView(st).private_cards.at(p); }, py::arg("player"))
      .def("public_card", [](const State& st) { return LeducView(st).public_card; })
      .def("round", [](const State& st) { return LeducView(st).round; })
      .def("money", [](const State& st) { return LeducView(st).money; })
      .def("pot", [](const State& st) { return LeducView(st).pot; })
      .def("round1", [](const State& st) { return LeducView(st).round1; })
      .def("round2", [](const State& st) { return LeducView(st).round2; })
      .def("move_number", &State::MoveNumber)
      .def("num_players", &State::NumPlayers)
      .def("get_game", [](const State& s) { return std::const_pointer_cast<Game>(s.GetGame()); });

  py::class_<BatchedState>(m, "BatchedState")
      .def("__len__", &BatchedState::size)
      .def("legal_actions_mask_bits",
           [](const BatchedState& b) {
             return as_array(b.LegalActionsMaskBits(), {b.size(), b.GetGame()->Desc().mask_words});
           })
      .def("apply_actions",
           [](BatchedState& b, py::array_t<int32_t, py::array::c_style | py::array::forcecast> a) {
             b.ApplyActions(std::vector<int32_t>(a.data(), a.data() + a.size()));
           },
           py::arg("actions"))
      .def("is_terminal", [](const BatchedState& b) { return as_array(b.IsTerminal(), {b.size()}); })
      .def("current_player", [](const BatchedState& b) { return as_array(b.CurrentPlayer(), {b.size()}); })
      .def("returns", [](const BatchedState& b) { return as_array(b.Returns(), {b.size(), b.GetGame()->NumPlayers()}); })
      .def("observation_tensor",
           [](const BatchedState& b, Player p) {
             return as_array(b.ObservationTensor(p), {b.size(), b.GetGame()->ObservationTensorSize()});
           },
           py::arg("player"))
      .def("information_state_tensor",
           [](const BatchedState& b, Player p) {
             return as_array(b.InformationStateTensor(p), {b.size(), b.GetGame()->InformationStateTensorSize()});
           },
           py::arg("player"))
      .def("clone", [](const BatchedState& b) { return BatchedState(b); });

  // bots.cc:106-111; here also subclassable from Python (evaluate(state) -> [value per player], prior(state) ->
  // [(action, probability)]): MCTSBot.mcts_search / step route the device search's requests to it
  py::class_<Evaluator, PyEvaluator, std::shared_ptr<Evaluator>>(m, "Evaluator")
      .def(py::init<>())
      .def("evaluate", &Evaluator::Evaluate, py::arg("state"))
      .def("prior", &Evaluator::Prior, py::arg("state"));
  py::class_<RandomRolloutEvaluator, Evaluator, std::shared_ptr<RandomRolloutEvaluator>>(m, "RandomRolloutEvaluator")
      .def(py::init<int, int>(), py::arg("n_rollouts"), py::arg("seed"))
      .def("evaluate", &RandomRolloutEvaluator::Evaluate, py::arg("state"))
      .def("evaluate_batch",
           [](RandomRolloutEvaluator& e, const BatchedState& b) {
             return as_array(e.EvaluateBatch(b), {b.size(), b.GetGame()->NumPlayers()});
           })
      .def("prior", &RandomRolloutEvaluator::Prior, py::arg("state"));

  py::class_<SearchNode>(m, "SearchNode")  // bots.cc:119-131
      .def_readonly("action", &SearchNode::action)
      .def_readonly("prior", &SearchNode::prior)
      .def_readonly("player", &SearchNode::player)
      .def_readonly("explore_count", &SearchNode::explore_count)
      .def_readonly("total_reward", &SearchNode::total_reward)
      .def_readonly("outcome", &SearchNode::outcome)
      .def_readonly("children", &SearchNode::children)
      .def("best_child", &SearchNode::BestChild)
      .def("to_string", &SearchNode::ToString, py::arg("state"))
      .def("children_str", &SearchNode::ChildrenStr, py::arg("state"));

  py::class_<Bot, PyBot>(m, "Bot")  // python/pybind11/bots.cc:57-98 (spiel_bots.h:73-185); Python bots derive from it
      .def(py::init<>())
      .def("step", &Bot::Step, py::arg("state"))
      .def("restart", &Bot::Restart)
      .def("restart_at", &Bot::RestartAt, py::arg("state"))
      .def("provides_force_action", &Bot::ProvidesForceAction)
      .def("force_action", &Bot::ForceAction, py::arg("state"), py::arg("action"))
      .def("inform_action", &Bot::InformAction, py::arg("state"), py::arg("player_id"), py::arg("action"))
      .def("inform_actions", &Bot::InformActions, py::arg("state"), py::arg("actions"))
      .def("provides_policy", &Bot::ProvidesPolicy)
      .def("get_policy", &Bot::GetPolicy, py::arg("state"))
      .def("step_with_policy", &Bot::StepWithPolicy, py::arg("state"))
      .def("is_clonable", &Bot::IsClonable);

  // bots.cc:177-196: one episode with a bot per player; the stock bots
  m.def("evaluate_bots", [](State* state, const std::vector<Bot*>& bots, int seed) { return EvaluateBots(state, bots, seed); },
        py::arg("state"), py::arg("bots"), py::arg("seed"));
  m.def("make_uniform_random_bot", [](Player player_id, int seed) { return MakeUniformRandomBot(player_id, seed); },
        py::arg("player_id"), py::arg("seed"));
  m.def("make_stateful_random_bot",
        [](std::shared_ptr<Game> game, Player player_id, int seed) { return MakeStatefulRandomBot(*game, player_id, seed); },
        py::arg("game"), py::arg("player_id"), py::arg("seed"));
  m.def("make_policy_bot",
        [](std::shared_ptr<Game> game, Player player_id, int seed, std::shared_ptr<Policy> policy) {
          return MakePolicyBot(*game, player_id, seed, std::move(policy));
        },
        py::arg("game"), py::arg("player_id"), py::arg("seed"), py::arg("policy"));

  py::enum_<ChildSelectionPolicy>(m, "ChildSelectionPolicy")  // bots.cc:113-117
      .value("UCT", ChildSelectionPolicy::UCT)
      .value("PUCT", ChildSelectionPolicy::PUCT);
  py::class_<MCTSBot, Bot>(m, "MCTSBot")  // bots.cc:133-149 (+ the constructor's remaining arguments, mcts.h:161-169)
      .def(py::init([](std::shared_ptr<Game> game, std::shared_ptr<Evaluator> evaluator, double uct_c,
                       int max_simulations, int64_t max_memory_mb, bool solve, int seed, bool verbose,
                       ChildSelectionPolicy policy, double max_wall_clock_time, double dirichlet_alpha,
                       double dirichlet_epsilon, bool dont_return_chance_node) {
             return new MCTSBot(*game, std::move(evaluator), uct_c, max_simulations, max_memory_mb, solve, seed, verbose,
                                policy, dirichlet_alpha, dirichlet_epsilon, dont_return_chance_node, max_wall_clock_time);
           }),
           py::arg("game"), py::arg("evaluator"), py::arg("uct_c"), py::arg("max_simulations"),
           py::arg("max_memory_mb"), py::arg("solve"), py::arg("seed"), py::arg("verbose"),
           py::arg("child_selection_policy") = ChildSelectionPolicy::UCT, py::arg("max_wall_clock_time") = -1.0,
           py::arg("dirichlet_alpha") = 0.0, py::arg("dirichlet_epsilon") = 0.0,
           py::arg("dont_return_chance_node") = false)
      // bots.cc:147-148 releases the GIL around step: here too whenever the evaluator is the C++
      // RandomRolloutEvaluator; a Python Evaluator is called back from inside the search and keeps it
      .def("step",
           [](MCTSBot& bot, const State& state) {
             if (!bot.EvaluatorIsNative()) return bot.Step(state);
             py::gil_scoped_release release;
             return bot.Step(state);
           },
           py::arg("state"))
      .def("step_with_policy", &MCTSBot::StepWithPolicy, py::arg("state"))  // spiel_bots.h:105-112
      .def("mcts_search",
           [](MCTSBot& bot, const State& state) {
             if (!bot.EvaluatorIsNative()) return bot.MCTSearch(state);
             py::gil_scoped_release release;
             return bot.MCTSearch(state);
           },
           py::arg("state"))
      .def("step_batch", &MCTSBot::StepBatch, py::arg("states"), py::call_guard<py::gil_scoped_release>());

  // python/pybind11/policy.cc:90-222: Policy, TabularPolicy, UniformPolicy, PreferredActionPolicy and the factories.
  // Python subclasses of Policy (action_probabilities(state, player) -> {action: prob}) are accepted by the judge
  // functions through the trampoline.
  py::class_<Policy, PyPolicy, std::shared_ptr<Policy>>(m, "Policy")
      .def(py::init<>())
      .def("action_probabilities", [](const Policy& p, const State& s) { return p.GetStatePolicyAsMap(s); }, py::arg("state"))
      .def("action_probabilities",
           [](const Policy& p, const State& s, Player pl) {
             std::unordered_map<Action, double> m;
             for (const auto& ap : p.GetStatePolicy(s, pl)) m[ap.first] = ap.second;
             return m;
           },
           py::arg("state"), py::arg("player_id"))
      .def("get_state_policy", [](const Policy& p, const State& s) { return p.GetStatePolicy(s); }, py::arg("state"))
      .def("get_state_policy", [](const Policy& p, const State& s, Player pl) { return p.GetStatePolicy(s, pl); },
           py::arg("state"), py::arg("player"))
      .def("get_state_policy", [](const Policy& p, const std::string& k) { return p.GetStatePolicy(k); }, py::arg("info_state"))
      .def("get_state_policy_as_map", [](const Policy& p, const std::string& k) { return p.GetStatePolicyAsMap(k); },
           py::arg("info_state"))
      .def("get_state_policy_as_parallel_vectors",
           [](const Policy& p, const State& s) { return p.GetStatePolicyAsParallelVectors(s); }, py::arg("state"))
      .def("get_state_policy_as_parallel_vectors",
           [](const Policy& p, const std::string& k) { return p.GetStatePolicyAsParallelVectors(k); }, py::arg("info_state"));
  py::class_<TabularPolicy, Policy, std::shared_ptr<TabularPolicy>>(m, "TabularPolicy")
      .def(py::init<TabularPolicyTable>(), py::arg("table"))
      .def(py::init([](std::shared_ptr<Game> g) { return std::make_shared<TabularPolicy>(*g); }), py::arg("game"))
      .def("policy_table", [](const TabularPolicy& p) { return p.PolicyTable(); })
      .def("set_prob", &TabularPolicy::SetProb, py::arg("info_state"), py::arg("action"), py::arg("prob"))
      .def("set_state_policy", &TabularPolicy::SetStatePolicy, py::arg("info_state"), py::arg("state_policy"))
      .def("size", &TabularPolicy::size)
      .def("__len__", &TabularPolicy::size)
      .def("to_string", &TabularPolicy::ToString)
      .def("__str__", &TabularPolicy::ToString)
      .def("__repr__", &TabularPolicy::ToString);
  py::class_<UniformPolicy, Policy, std::shared_ptr<UniformPolicy>>(m, "UniformPolicy").def(py::init<>());
  py::class_<PreferredActionPolicy, Policy, std::shared_ptr<PreferredActionPolicy>>(m, "PreferredActionPolicy")
      .def(py::init<std::vector<Action>>(), py::arg("preference_order"));
  m.def("UniformRandomPolicy", [](std::shared_ptr<Game> g) { return GetUniformPolicy(*g); }, py::arg("game"));
  m.def("GetFirstActionPolicy", [](std::shared_ptr<Game> g) { return GetFirstActionPolicy(*g); }, py::arg("game"));
  m.def("GetEmptyTabularPolicy", [](std::shared_ptr<Game> g, bool uniform) { return GetEmptyTabularPolicy(*g, uniform); },
        py::arg("game"), py::arg("initialize_to_uniform") = false);
  m.def("ToTabularPolicy", [](std::shared_ptr<Game> g, const Policy& p) { return TabularPolicy(*g, p); }, py::arg("game"),
        py::arg("policy"));

  py::class_<CFRInfoStateValues>(m, "CFRInfoStateValues")
      .def_readonly("legal_actions", &CFRInfoStateValues::legal_actions)
      .def_readonly("cumulative_regrets", &CFRInfoStateValues::cumulative_regrets)
      .def_readonly("cumulative_policy", &CFRInfoStateValues::cumulative_policy)
      .def_readonly("current_policy", &CFRInfoStateValues::current_policy);

  py::class_<CFRSolverBase>(m, "CFRSolverBase")
      .def("evaluate_and_update_policy", py::overload_cast<>(&CFRSolverBase::EvaluateAndUpdatePolicy))
      .def("evaluate_and_update_policy", py::overload_cast<int>(&CFRSolverBase::EvaluateAndUpdatePolicy),
           py::arg("iterations"))
      .def("average_policy", [](const CFRSolverBase& s) { return std::make_shared<TabularPolicy>(s.TabularAveragePolicy()); })
      .def("tabular_average_policy", [](const CFRSolverBase& s) { return std::make_shared<TabularPolicy>(s.TabularAveragePolicy()); })
      .def("current_policy", [](const CFRSolverBase& s) { return std::make_shared<TabularPolicy>(s.TabularCurrentPolicy()); })
      .def("info_state_values_table", &CFRSolverBase::InfoStateValuesTable)
      .def("serialize", &CFRSolverBase::Serialize, py::arg("double_precision") = -1, py::arg("delimiter") = "<~>");
  py::class_<CFRSolver, CFRSolverBase>(m, "CFRSolver")  // policy.cc:224-262 (pickle = serialize / deserialize)
      .def(py::init([](std::shared_ptr<Game> g) { return new CFRSolver(*g); }), py::arg("game"))
      .def(py::pickle([](const CFRSolver& s) { return s.Serialize(); },
                      [](const std::string& t) { return DeserializeCFRSolver(t); }));
  py::class_<CFRPlusSolver, CFRSolverBase>(m, "CFRPlusSolver")
      .def(py::init([](std::shared_ptr<Game> g) { return new CFRPlusSolver(*g); }), py::arg("game"))
      .def(py::pickle([](const CFRPlusSolver& s) { return s.Serialize(); },
                      [](const std::string& t) { return DeserializeCFRPlusSolver(t); }));
  py::class_<CFRBRSolver, CFRSolverBase>(m, "CFRBRSolver")  // policy.cc:264-298
      .def(py::init([](std::shared_ptr<Game> g) { return new CFRBRSolver(*g); }), py::arg("game"))
      .def("evaluate_and_update_policy", py::overload_cast<>(&CFRBRSolver::EvaluateAndUpdatePolicy))
      .def("evaluate_and_update_policy", py::overload_cast<int>(&CFRBRSolver::EvaluateAndUpdatePolicy),
           py::arg("iterations"))
      .def(py::pickle([](const CFRBRSolver& s) { return s.Serialize(); },
                      [](const std::string& t) { return DeserializeCFRBRSolver(t); }));
  m.def("deserialize_cfr_br_solver", [](const std::string& t) { return DeserializeCFRBRSolver(t); });
  m.def("deserialize_cfr_solver", [](const std::string& t) { return DeserializeCFRSolver(t); });
  m.def("deserialize_cfr_plus_solver", [](const std::string& t) { return DeserializeCFRPlusSolver(t); });

  m.def("serialize_game_and_state",
        [](std::shared_ptr<Game> g, const State& s) { return SerializeGameAndState(*g, s); }, py::arg("game"),
        py::arg("state"));
  m.def("deserialize_game_and_state", [](const std::string& t) {  // pyspiel.cc:733-741
    auto gs = DeserializeGameAndState(t);
    return std::make_pair(std::const_pointer_cast<Game>(gs.first), std::move(gs.second));
  });

  py::class_<TabularBestResponse>(m, "TabularBestResponse")  // python/pybind11/policy.cc:138-162
      .def(py::init([](std::shared_ptr<Game> g, int responder, const Policy& p) { return new TabularBestResponse(*g, responder, &p); }),
           py::arg("game"), py::arg("best_responder"), py::arg("policy"))
      .def(py::init([](std::shared_ptr<Game> g, int responder, const TabularPolicyTable& t) {
             return new TabularBestResponse(*g, responder, t);
           }),
           py::arg("game"), py::arg("best_responder"), py::arg("policy_table"))
      .def("value", [](TabularBestResponse& br, const std::string& /*history: the root*/) { return br.Value(); }, py::arg("history") = "")
      .def("value_from_state", [](TabularBestResponse& br, const State& s) {
             if (s.MoveNumber() != 0) SpielFatalError("the device best response reports the value at the root");
             return br.Value();
           }, py::arg("state"))
      .def("get_best_response_policy", &TabularBestResponse::GetBestResponsePolicy)
      .def("get_best_response_actions", &TabularBestResponse::GetBestResponseActions)
      .def("set_policy", [](TabularBestResponse& br, const Policy& p) { br.SetPolicy(&p); }, py::arg("policy"))
      .def("set_policy", [](TabularBestResponse& br, const TabularPolicyTable& t) { br.SetPolicy(t); }, py::arg("policy_table"));

  // pyspiel.exploitability / nash_conv / expected_returns (python/pybind11/policy.cc) for tabular policies
  m.def("exploitability", [](std::shared_ptr<Game> g, const Policy& p) { return Exploitability(*g, p); },
        py::arg("game"), py::arg("policy"));
  m.def("nash_conv", [](std::shared_ptr<Game> g, const Policy& p) { return NashConv(*g, p); },
        py::arg("game"), py::arg("policy"));
  m.def("expected_returns", [](std::shared_ptr<Game> g, const Policy& p) { return ExpectedReturns(*g, p); },
        py::arg("game"), py::arg("policy"));

  // pyspiel.kuhn_poker.get_optimal_policy (python/pybind11/games_kuhn_poker.cc:23-24)
  py::module_ kuhn = m.def_submodule("kuhn_poker");
  kuhn.def("get_optimal_policy", [](double alpha) { return std::make_shared<TabularPolicy>(kuhn_poker::GetOptimalPolicy(alpha)); },
           py::arg("alpha"));

  // ---- python/pybind11/observer.cc:30-97 ----
  py::enum_<PrivateInfoType>(m, "PrivateInfoType")
      .value("NONE", PrivateInfoType::kNone)
      .value("SINGLE_PLAYER", PrivateInfoType::kSinglePlayer)
      .value("ALL_PLAYERS", PrivateInfoType::kAllPlayers);
  py::class_<IIGObservationType>(m, "IIGObservationType")
      .def(py::init([](bool public_info, bool perfect_recall, PrivateInfoType private_info) {
             return IIGObservationType{public_info, perfect_recall, private_info};
           }),
           py::arg("public_info") = true, py::arg("perfect_recall") = false,
           py::arg("private_info") = PrivateInfoType::kSinglePlayer)
      .def_readonly("public_info", &IIGObservationType::public_info)
      .def_readonly("perfect_recall", &IIGObservationType::perfect_recall)
      .def_readonly("private_info", &IIGObservationType::private_info)
      .def("__eq__", [](const IIGObservationType& a, const IIGObservationType& b) { return a == b; });
  py::class_<Observer, std::shared_ptr<Observer>>(m, "Observer")
      .def("__str__", [](const Observer&) { return "Observer()"; });
  py::class_<SpanTensorInfo>(m, "SpanTensorInfo")
      .def_property_readonly("name", [](const SpanTensorInfo& i) { return i.name(); })
      .def_property_readonly("shape", [](const SpanTensorInfo& i) { return i.vector_shape(); })
      .def("__str__", &SpanTensorInfo::DebugString);
  py::class_<SpanTensor>(m, "SpanTensor")
      .def_property_readonly("name", [](const SpanTensor& t) { return t.info().name(); })
      .def_property_readonly("shape", [](const SpanTensor& t) { return t.info().vector_shape(); })
      .def_property_readonly("data",
                             [](const SpanTensor& t) {  // zero-copy view into the Observation's buffer
                               std::vector<py::ssize_t> shape(t.info().shape().begin(), t.info().shape().end());
                               std::vector<py::ssize_t> strides(shape.size());
                               py::ssize_t stride = sizeof(float);
                               for (int i = static_cast<int>(shape.size()) - 1; i >= 0; --i) {
                                 strides[i] = stride;
                                 stride *= shape[i];
                               }
                               py::capsule keep(t.data(), [](void*) {});
                               return py::array_t<float>(shape, strides, t.data(), keep);
                             })
      .def("__str__", &SpanTensor::DebugString);
  py::class_<Observation>(m, "_Observation", py::buffer_protocol())
      .def(py::init([](std::shared_ptr<Game> game, std::shared_ptr<Observer> observer) {
             return new Observation(*game, std::move(observer));
           }),
           py::arg("game"), py::arg("observer"))
      .def("tensors",
           [](py::object self) {  // every view keeps the Observation (the owner of the buffer) alive
             py::list out;
             for (SpanTensor& t : self.cast<Observation&>().tensors()) {
               py::object e = py::cast(std::move(t));
               py::detail::keep_alive_impl(e, self);
               out.append(e);
             }
             return out;
           })
      .def("tensors_info", &Observation::tensors_info)
      .def("string_from", &Observation::StringFrom, py::arg("state"), py::arg("player"))
      .def("set_from", &Observation::SetFrom, py::arg("state"), py::arg("player"))
      .def("has_string", &Observation::HasString)
      .def("has_tensor", &Observation::HasTensor)
      // python/pybind11/observer.cc:88-92: compress() -> bytes, decompress(bytes)
      .def("compress", [](const Observation& o) { return py::bytes(o.Compress()); })
      .def("decompress", [](Observation& o, py::bytes b) { o.Decompress(static_cast<std::string>(b)); })
      .def_buffer([](Observation& o) -> py::buffer_info {
        return py::buffer_info(o.Tensor().data(), sizeof(float), py::format_descriptor<float>::format(), 1,
                               {o.Tensor().size()}, {sizeof(float)});
      });

  // ---- game submodules (python/pybind11/games_tic_tac_toe.cc:37-100, games_leduc_poker.cc:28-60,
  // games_connect_four.cc:47-95).  One State class serves every game here, so the game-specific accessors are
  // methods of State that refuse other games; the struct types (…StateStruct / …ActionStruct / …GameParams) are classes of the
  // submodules as in the reference.
  py::module_ ttt = m.def_submodule("tic_tac_toe");
  py::enum_<TttCellState>(ttt, "CellState")
      .value("EMPTY", TttCellState::kEmpty).value("NOUGHT", TttCellState::kNought).value("CROSS", TttCellState::kCross)
      .export_values();
  ttt.attr("NUM_ROWS") = py::int_(3);
  ttt.attr("NUM_COLS") = py::int_(3);
  ttt.attr("NUM_CELLS") = py::int_(9);
  ttt.def("player_to_cellstate", [](Player p) {  // tic_tac_toe.cc:52-63
    if (p == 0) return TttCellState::kCross;
    if (p == 1) return TttCellState::kNought;
    SpielFatalError("Invalid player id " + std::to_string(p));
  });
  ttt.def("cellstate_to_string", [](TttCellState c) {  // tic_tac_toe.cc:65-77
    return std::string(c == TttCellState::kEmpty ? "." : c == TttCellState::kNought ? "o" : "x");
  });
  bind_spiel_struct<tic_tac_toe::TicTacToeStateStruct, StateStruct>(ttt, "TicTacToeStateStruct")
      .def_readwrite("current_player", &tic_tac_toe::TicTacToeStateStruct::current_player)
      .def_readwrite("board", &tic_tac_toe::TicTacToeStateStruct::board);
  bind_spiel_struct<tic_tac_toe::TicTacToeObservationStruct, ObservationStruct>(ttt, "TicTacToeObservationStruct")
      .def_readwrite("current_player", &tic_tac_toe::TicTacToeObservationStruct::current_player)
      .def_readwrite("board", &tic_tac_toe::TicTacToeObservationStruct::board);
  bind_spiel_struct<tic_tac_toe::TicTacToeActionStruct, ActionStruct>(ttt, "TicTacToeActionStruct")
      .def_readwrite("row", &tic_tac_toe::TicTacToeActionStruct::row)
      .def_readwrite("col", &tic_tac_toe::TicTacToeActionStruct::col);
  py::module_ c4 = m.def_submodule("connect_four");
  c4.attr("__doc__") = "connect_four: states and games are pyspiel_hip.State / pyspiel_hip.Game (pickle included); the struct "
                       "types of games_connect_four.cc:47-95";
  bind_spiel_struct<connect_four::ConnectFourStateStruct, StateStruct>(c4, "ConnectFourStateStruct")
      .def_readwrite("current_player", &connect_four::ConnectFourStateStruct::current_player)
      .def_readwrite("board", &connect_four::ConnectFourStateStruct::board)
      .def_readwrite("is_terminal", &connect_four::ConnectFourStateStruct::is_terminal)
      .def_readwrite("winner", &connect_four::ConnectFourStateStruct::winner);
  bind_spiel_struct<connect_four::ConnectFourObservationStruct, ObservationStruct>(c4, "ConnectFourObservationStruct")
      .def_readwrite("current_player", &connect_four::ConnectFourObservationStruct::current_player)
      .def_readwrite("board", &connect_four::ConnectFourObservationStruct::board)
      .def_readwrite("is_terminal", &connect_four::ConnectFourObservationStruct::is_terminal)
      .def_readwrite("winner", &connect_four::ConnectFourObservationStruct::winner);
  bind_spiel_struct<connect_four::ConnectFourActionStruct, ActionStruct>(c4, "ConnectFourActionStruct")
      .def_readwrite("column", &connect_four::ConnectFourActionStruct::column);
  bind_spiel_struct<connect_four::ConnectFourGameParams, GameParametersStruct>(c4, "ConnectFourGameParams")
      .def_readwrite("rows", &connect_four::ConnectFourGameParams::rows)
      .def_readwrite("columns", &connect_four::ConnectFourGameParams::columns)
      .def_readwrite("x_in_row", &connect_four::ConnectFourGameParams::x_in_row)
      .def_readwrite("egocentric_obs_tensor", &connect_four::ConnectFourGameParams::egocentric_obs_tensor);
  py::module_ leduc = m.def_submodule("leduc_poker");
  leduc.attr("INVALID_CARD") = py::int_(-10000);  // leduc_poker.h:57 kInvalidCard
  py::enum_<LeducActionType>(leduc, "ActionType")
      .value("FOLD", LeducActionType::kFold).value("CALL", LeducActionType::kCall).value("RAISE", LeducActionType::kRaise)
      .export_values();

  py::enum_<AverageType>(m, "MCCFRAverageType").value("SIMPLE", AverageType::kSimple).value("FULL", AverageType::kFull);
  py::class_<ExternalSamplingMCCFRSolver>(m, "ExternalSamplingMCCFRSolver")  // policy.cc:300-333
      .def(py::init([](std::shared_ptr<Game> g, int seed, AverageType t) {
             return new ExternalSamplingMCCFRSolver(*g, seed, t);
           }),
           py::arg("game"), py::arg("seed") = 0, py::arg("avg_type") = AverageType::kSimple)
      .def("run_iteration", py::overload_cast<>(&ExternalSamplingMCCFRSolver::RunIteration))
      // RunIteration(std::mt19937*): the generator lives in the returned object; seeded like the reference's
      // ExternalSamplingMCCFRSolver(game, seed, ...) it reproduces that solver's draws one for one
      .def("run_iterations_mt19937",
           [](ExternalSamplingMCCFRSolver& s, uint32_t seed, int iterations) {
             std::mt19937 rng(seed);
             for (int i = 0; i < iterations; ++i) s.RunIteration(&rng);
           },
           py::arg("seed"), py::arg("iterations"))
      .def("run_mini_batch", &ExternalSamplingMCCFRSolver::RunMiniBatch, py::arg("trajectories"))
      .def("average_policy",
           [](const ExternalSamplingMCCFRSolver& s) { return std::make_shared<TabularPolicy>(s.TabularAveragePolicy()); })
      .def("info_state_values_table", &ExternalSamplingMCCFRSolver::InfoStateValuesTable)
      .def("serialize", &ExternalSamplingMCCFRSolver::Serialize, py::arg("double_precision") = -1,
           py::arg("delimiter") = "<~>")
      .def(py::pickle([](const ExternalSamplingMCCFRSolver& s) { return s.Serialize(); },  // policy.cc:322-333
                      [](const std::string& t) { return DeserializeExternalSamplingMCCFRSolver(t); }));
  m.def("deserialize_external_sampling_mccfr_solver",
        [](const std::string& t) { return DeserializeExternalSamplingMCCFRSolver(t); });
  py::class_<OutcomeSamplingMCCFRSolver>(m, "OutcomeSamplingMCCFRSolver")  // policy.cc:334-370
      .def(py::init([](std::shared_ptr<Game> g, double epsilon, int seed) {
             return new OutcomeSamplingMCCFRSolver(*g, epsilon, seed);
           }),
           py::arg("game"), py::arg("epsilon") = OutcomeSamplingMCCFRSolver::kDefaultEpsilon, py::arg("seed") = -1)
      .def("run_iteration", static_cast<void (OutcomeSamplingMCCFRSolver::*)()>(&OutcomeSamplingMCCFRSolver::RunIteration))
      .def("run_mini_batch", &OutcomeSamplingMCCFRSolver::RunMiniBatch, py::arg("episodes"))
      .def("average_policy",
           [](const OutcomeSamplingMCCFRSolver& s) { return std::make_shared<TabularPolicy>(s.TabularAveragePolicy()); })
      .def("info_state_values_table", &OutcomeSamplingMCCFRSolver::InfoStateValuesTable)
      .def("serialize", &OutcomeSamplingMCCFRSolver::Serialize, py::arg("double_precision") = -1,
           py::arg("delimiter") = "<~>")
      .def(py::pickle([](const OutcomeSamplingMCCFRSolver& s) { return s.Serialize(); },  // policy.cc:359-370
                      [](const std::string& t) { return DeserializeOutcomeSamplingMCCFRSolver(t); }));
  m.def("deserialize_outcome_sampling_mccfr_solver",
        [](const std::string& t) { return DeserializeOutcomeSamplingMCCFRSolver(t); });
}
