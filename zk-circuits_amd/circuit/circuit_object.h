// The object behind qpc_circuit handles, shared by circuits.cc and recursion.cc (not part of the ABI).
#pragma once
#include "builder.h"
struct qpc_circuit {
  uint32_t kind; qpc::Builder b;
  void* aux = nullptr; void (*aux_del)(void*) = nullptr;       // per-kind extra state (the aggregation circuit keeps its inner circuit's shape)
  qpc_circuit(uint32_t k, bool zk) : kind(k), b(zk) {}
  ~qpc_circuit() { if (aux && aux_del) aux_del(aux); }
  qpc_circuit(const qpc_circuit&) = delete; qpc_circuit& operator=(const qpc_circuit&) = delete;
};
void qpc_set_error(const char* m);                             // sets what qpc_last_error returns on this thread
