// The caller-owned tuning (include/fnssl.h: fnssl_tuning) as the kernels' host code sees it.  The library reads no
// environment variable: every knob comes from the process default (fnssl_tuning_set) or from the descriptor of the
// call in progress (TuningScope).  Timing-ablation knobs exist only in `make ABLATE=1` builds and stay there.
#pragma once

#include "../../include/fnssl.h"

namespace fnssl {

const fnssl_tuning& tuning();   // the call's (innermost TuningScope) or else this thread's snapshot of the process default

// knob value, 0 = default
inline int tune(int index) { return tuning().knob[index]; }
// ... accepted only inside [lo, hi] (anything else counts as "not set")
inline int tune(int index, int lo, int hi) {
  const int v = tune(index);
  return v >= lo && v <= hi ? v : 0;
}

// For the duration of one ABI call: use the descriptor's tuning (no-op when NULL).
struct TuningScope {
  explicit TuningScope(const fnssl_tuning* t);
  ~TuningScope();
  TuningScope(const TuningScope&) = delete;
  TuningScope& operator=(const TuningScope&) = delete;
  const fnssl_tuning* prev_;
  bool active_;
};

}  // namespace fnssl
