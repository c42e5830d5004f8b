/* Test hooks of libqpcircuit.so (not part of the ABI of include/qpcircuit.h). */
#ifndef QPCIRCUIT_INTERNAL_H
#define QPCIRCUIT_INTERNAL_H
#include "qpcircuit.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Builder fuzzing: a circuit of `num_ops` random builder operations (add / sub / mul / mul_const / mul_add / select / is_equal / not /
 * and / or / split_le + recombination / range_check / hash / permute_swap / connect of equal values / constants) over `num_inputs` input
 * targets, drawn from `seed`, with every intermediate value evaluated natively next to the builder calls. All `num_outputs` sampled
 * intermediate targets are registered as public inputs. Returns the circuit, the input values used (inputs[num_inputs]) and the natively
 * computed values of the public inputs (expected[*num_outputs], at most cap). The witness of these inputs must reproduce `expected`. */
int qpc_test_random_circuit(uint64_t seed, uint32_t zero_knowledge, uint32_t num_inputs, uint32_t num_ops, qpc_circuit** out, uint64_t* inputs,
                            uint64_t* expected, size_t cap, size_t* num_outputs);

#ifdef __cplusplus
}
#endif
#endif
