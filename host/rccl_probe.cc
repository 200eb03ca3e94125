// rccl_probe.cc -- RcclExchange with one rank, nothing else in the process: set-up, one all-reduce, tear-down.
#include <cstdio>
#include <vector>
#include "mlmcpi/exchange.hh"
#include "mlmcpi_hip.h"
int main() {
  mlmcpi::RcclExchange ex(0, 1, "/tmp/mlmcpi_probe_id", 0);
  std::vector<double> v = {1.5, -2.0};
  ex.allreduce_sum(v.data(), v.size());
  std::printf("RcclExchange ok: %g %g\n", v[0], v[1]);
  return 0;
}
