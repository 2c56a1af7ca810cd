// Action sampling shared by the rollout, random-step, MCTS and MCCFR kernels.
#ifndef OSG_SAMPLE_H_
#define OSG_SAMPLE_H_

#include "osg_common.h"

namespace osg {

// Draw a legal action like the oracle: chance nodes by SampleAction's CDF scan
// over ChanceOutcomes() with z = rng.unit() (spiel.cc:372-409), decision nodes
// uniformly over LegalActions() with rng.below(count) (mcts.cc:51-55).
template <class G, int W>
OSG_D int sample_action_w(const typename G::Params& p, const typename G::State& s, const MaskT<W>& m, int cur, Rng& rng) {
  if (cur == kChancePlayer) {
    int cnt = m.count();
    double z = rng.unit();  // drawn even for a single outcome, as the reference's call sites do
    double acc = 0.0;
    int last = -1;
    for (int k = 0; k < cnt; ++k) {
      int o = select_action(m, k);
      double pr = G::chance_prob(p, s, o);
      if (acc <= z && z < acc + pr) return o;
      acc += pr;
      last = o;
    }
    return last;  // unreachable for a valid distribution
  }
  return select_action(m, static_cast<int>(rng.below(static_cast<uint32_t>(m.count()))));
}

template <class G, int W>
OSG_D int sample_action(const typename G::Params& p, const typename G::State& s, const MaskT<W>& m, int cur, Rng& rng) {
  return sample_action_w<G, W>(p, s, m, cur, rng);
}

// Chance-node draw only (MCTS tree policy at chance nodes, mcts.cc:311-322).
template <class G, int W>
OSG_D int sample_action_chance(const typename G::Params& p, const typename G::State& s, const MaskT<W>& m, Rng& rng) {
  return sample_action_w<G, W>(p, s, m, kChancePlayer, rng);
}

}  // namespace osg
#endif  // OSG_SAMPLE_H_
