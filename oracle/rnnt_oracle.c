/*
 * rnnt_oracle.c — CPU restatement of the jakepoz/rnnt joint + transducer-loss hot path.
 *
 * *** TEST INFRASTRUCTURE ONLY. ***  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library, and only as the checker / the CPU
 * number reported beside the GPU one.  Nothing under rnnt_amd/ imports or links it.
 *
 * What it restates (reference paths relative to /root/reference):
 *   rnnt/joint.py:32-39   broadcast add -> tanh -> Linear(H->V)          (in-repo Python)
 *   rnnt/model.py:35-41   torchaudio.functional.rnnt_loss(..., blank=-1, clamp=-1,
 *                         reduction="mean")                               (THIRD PARTY)
 *   rnnt/train.py:133-134 loss.backward(): autograd of the two above
 *
 * Third-party provenance: the loss arithmetic lives in `torchaudio`, which is NOT under
 * /root/reference, is not installed in this image and is unpinned by the reference
 * (requirements.txt lists neither torch nor torchaudio).  The recurrences and the fused
 * log-softmax gradient are restated from the published algorithm (Graves 2012,
 * "Sequence Transduction with Recurrent Neural Networks") in the form SURVEY.md §8c
 * records.  The reference's own tests hold NO golden vectors for this path,
 * so the LOSS half is pinned by (a) brute-force alignment enumeration
 * (oracle/brute_force.py), (b) torch autograd through an independent log-space alpha
 * recursion (oracle/torch_check.py) and (c) the PUBLISHED known-answer vectors of the
 * third-party algorithm — the warp-transducer unit-test cases torchaudio's own tests reuse
 * to pin rnnt_loss (tests/golden/published_transducer_kat.json: costs and gradients,
 * reproduced to 3e-7; third-party published data, transcribed, not reference-held).
 * torchaudio itself cannot be run here: "parity unpinned" against torchaudio itself stands.
 * The JOINT half IS pinned: the .npz files under tests/golden hold logits and autograd gradients produced
 * by importing the reference's own rnnt.joint.JointNetwork (tests/golden/make_golden.py).
 *
 * Two instantiations: _f64 (ground truth) and _f32 (the reference's arithmetic type; also
 * the cpu_baseline "port").
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define REAL double
#define SUFFIX f64
#include "rnnt_oracle_body.inc"
#undef REAL
#undef SUFFIX

#define REAL float
#define SUFFIX f32
#include "rnnt_oracle_body.inc"
#undef REAL
#undef SUFFIX

int rnnt_oracle_version(void) { return 1; }
