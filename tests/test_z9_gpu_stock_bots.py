"""The stock bots of spiel_bots.h (MakeUniformRandomBot, MakeStatefulRandomBot, MakePolicyBot,
MakeFixedActionPreferenceBot) and EvaluateBots on the host mirror, in the shape of the reference's
algorithms/evaluate_bots_test.cc and examples/mcts_example.cc — a short version for every run; the reference's own
evaluate_bots_test.cc (200 000 episodes) is the OSG_SLOW_TESTS case of tests/test_z6_gpu_reference_tests_on_mirror.py."""
import json
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROGRAM = r'''
#include <cstdio>
#include <memory>
#include <vector>
#include "open_spiel/algorithms/evaluate_bots.h"
#include "open_spiel/algorithms/mcts.h"
#include "open_spiel/policy.h"
#include "open_spiel/spiel.h"
#include "open_spiel/spiel_bots.h"
using namespace open_spiel;

int main() {
  // evaluate_bots_test.cc:25-45 with fewer episodes: uniform random vs a bot that follows the game in its own state
  {
    auto game = LoadGame("kuhn_poker");
    auto bot0 = MakeUniformRandomBot(0, /*seed=*/1234);
    auto bot1 = MakeStatefulRandomBot(*game, 1, /*seed=*/4321);
    double sum0 = 0, sum1 = 0;
    const int episodes = 4000;
    for (int it = 0; it < episodes; ++it) {
      auto r = EvaluateBots(game->NewInitialState().get(), {bot0.get(), bot1.get()}, /*seed=*/it);
      sum0 += r[0]; sum1 += r[1];
    }
    std::printf("{\"kuhn_random_vs_stateful\": [%.6f, %.6f]}\n", sum0 / episodes, sum1 / episodes);
  }
  // evaluate_bots_test.cc:47-70: against a PolicyBot over the uniform TabularPolicy
  {
    auto game = LoadGame("kuhn_poker");
    auto bot0 = MakeUniformRandomBot(0, 1234);
    std::unique_ptr<Policy> uniform = std::make_unique<TabularPolicy>(GetUniformPolicy(*game));
    auto bot1 = MakePolicyBot(*game, Player{1}, 4321, std::move(uniform));
    double sum0 = 0;
    const int episodes = 4000;
    for (int it = 0; it < episodes; ++it) sum0 += EvaluateBots(game->NewInitialState().get(), {bot0.get(), bot1.get()}, it)[0];
    std::printf("{\"kuhn_random_vs_policy_bot\": %.6f}\n", sum0 / episodes);
  }
  // examples/mcts_example.cc in small: MCTSBot (the device search) against a uniform random bot, both seats
  {
    auto game = LoadGame("tic_tac_toe");
    auto evaluator = std::make_shared<algorithms::RandomRolloutEvaluator>(/*n_rollouts=*/8, /*seed=*/5);
    int mcts_losses = 0, games = 0;
    double mcts_total = 0;
    for (int seat = 0; seat < 2; ++seat) {
      for (int g = 0; g < 6; ++g) {
        algorithms::MCTSBot mcts(*game, evaluator, /*uct_c=*/2.0, /*max_simulations=*/300, /*max_memory_mb=*/10, /*solve=*/true,
                                 /*seed=*/100 + g, /*verbose=*/false);
        auto random = MakeUniformRandomBot(1 - seat, 900 + g);
        std::vector<Bot*> bots(2);
        bots[seat] = &mcts; bots[1 - seat] = random.get();
        auto r = EvaluateBots(game->NewInitialState().get(), bots, g);
        mcts_total += r[seat];
        mcts_losses += r[seat] < 0;
        ++games;
      }
    }
    std::printf("{\"ttt_mcts_vs_random\": {\"games\": %d, \"mcts_losses\": %d, \"mcts_mean_return\": %.4f}}\n", games, mcts_losses,
                mcts_total / games);
  }
  // a preference bot plays its first legal preference
  {
    auto game = LoadGame("connect_four");
    auto state = game->NewInitialState();
    auto pref = MakeFixedActionPreferenceBot(0, {9, 3, 0});
    std::printf("{\"preference_bot_action\": %d}\n", static_cast<int>(pref->Step(*state)));
  }
  return 0;
}
'''


def test_stock_bots_and_evaluate_bots_on_the_mirror(tmp_path):
    import __graft_entry__ as ge
    ge.build()
    src = tmp_path / "bots.cc"
    src.write_text(PROGRAM)
    exe = tmp_path / "bots"
    lib_dir = os.path.join(ROOT, "open_spiel_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-w", "-I", os.path.join(ROOT, "include"), "-I", ROOT, str(src), "-o", str(exe),
                           "-L", lib_dir, "-losg_hip", f"-Wl,-rpath,{lib_dir}"])
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    out = {}
    for ln in r.stdout.splitlines():
        if ln.startswith("{"):
            out.update(json.loads(ln))
    print(out)
    a, b = out["kuhn_random_vs_stateful"]
    assert abs(a - 0.125) < 0.07 and abs(b + 0.125) < 0.07       # evaluate_bots_test.cc:43-44 has 0.01 at 100 000 episodes
    assert abs(out["kuhn_random_vs_policy_bot"] - 0.125) < 0.07
    m = out["ttt_mcts_vs_random"]
    assert m["games"] == 12 and m["mcts_losses"] == 0 and m["mcts_mean_return"] > 0.5
    assert out["preference_bot_action"] == 3                      # 9 is not a column; 3 is the first legal preference
