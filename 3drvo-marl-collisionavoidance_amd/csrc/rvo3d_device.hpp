// rvo3d_device.hpp -- gfx950 device code of the batched 3D-RVO drone step.
//
// One thread = one drone; a workgroup holds EPB whole environments so the
// whole step (RVO reward sweep -> integrate -> observation sweep -> optional
// auto-reset + re-observe -> outputs) is ONE launch with workgroup barriers
// between the phases.  Neighbour position / velocity / radius / priority are
// staged in LDS (64 B per drone).
//
// Cost structure (fp64 VALU is the scarce resource):
//   * stage G (packed fp32, full lane utilisation): a CONSERVATIVE candidate
//     filter over all neighbours - in range (|dp|^2 <= 100 + band), approaching
//     (v.rel > -eps) or nearly touching - builds a 64-bit candidate mask per
//     lane.  The bands bound the fp32 error (host-computed from the map size),
//     so a pair the reference would act on is never dropped;
//   * stage X (fp64, candidates only) repeats every test exactly: squared norms
//     against T(tau) = max{x : sqrt(x) <= tau} (equivalent to the reference's
//     `norm(..) <= tau` for a correctly rounded sqrt), collision, v.rel, then a
//     conservative cone pre-filter (beta >= alpha + 1e-4 rad, no asin/acos/div)
//     and only for the rest asin / acos / the TTC quadratic;
//   * outputs are quantised without fp64 divisions where that is provably exact
//     after the float32 cast, and observations are written once, coalesced.
//
// Decision arithmetic is fp64 and follows the reference's evaluation order
// (compile with -ffp-contract=off; the explicit __builtin_fma calls model the
// OpenBLAS ddot the reference goes through, see DESIGN.md "arithmetic model").
// Reference citations are relative to the reference root.
#pragma once
//
// Files: rvo3d_params.hpp (parameter blocks), rvo3d_math.hpp (arithmetic model, per-drone
// pieces), rvo3d_lds.hpp (LDS views), rvo3d_pairs.hpp (pair pipeline), rvo3d_step.hpp (the step
// kernel), rvo3d_aux_kernels.hpp (resets, tables, classical RVO selection), rvo3d_rollout_kernels.hpp (the
// trainer's per-step glue: policy heads + sampling, episode bookkeeping), rvo3d_policy_mlp.hpp (config 3's MLP(256, 256)
// policy step on the matrix cores).
#pragma once

#include "rvo3d_params.hpp"
#include "rvo3d_math.hpp"
#include "rvo3d_lds.hpp"
#include "rvo3d_pairs.hpp"
#include "rvo3d_step.hpp"
#include "rvo3d_aux_kernels.hpp"
#include "rvo3d_rollout_kernels.hpp"
#include "rvo3d_policy_mlp.hpp"
