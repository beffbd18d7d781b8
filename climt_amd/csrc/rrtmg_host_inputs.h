// rrtmg_host_inputs.h -- the inputs of a host-pointer (memspace = 0) call: what has to cross PCIe, and what does not.
//
// climt hands every input over as a full [layer][column] array (lw/component.py:449-516, sw/component.py:594-662) -- the
// well-mixed gases (CO2, CH4, N2O, O2, the CFCs) included, whose arrays hold ONE number in every model that does not carry
// them as tracers, and cloud / aerosol arrays that are all zeros in a model without clouds or aerosols.  At 8192 columns x 60
// layers that is 4 MB per array and 55-63 MB per band array; PCIe moves it at 56 GB/s, the GPU fills it at 4 TB/s.  So:
//   * an array of >= kScanMin doubles whose head is uniform is scanned on a few persistent host threads (memory rate, in the
//     background while the arrays that certainly differ go up); if EVERY element has the bits of the first one, the device
//     buffer is filled by a kernel instead of uploaded -- and not even that when the buffer still holds the same fill from the
//     previous call (rrtmg::DevBuf::uniform);
//   * the arrays for which zeros mean "nothing to add" (band optical depths given directly) are not materialised at all when
//     they are entirely +0.0: the device code takes its "array absent" path, which adds the same +0.0;
//   * unit factors the caller would otherwise apply on the host -- Pa -> mbar, kg m^-2 -> g m^-2, the water-vapour mass ->
//     volume mixing ratio -- are applied on the device after the upload (rrtmg_{sw,lw}_args::*_scale), with the operations
//     numpy would have used (one rounding per product, contraction off): same bits, no host pass over the array.
// Device-pointer calls (memspace = 1) pass through untouched.
#pragma once
#include <cstddef>
#include <vector>

#include <hip/hip_runtime.h>

struct rrtmg_ctx;

namespace rrtmg {

constexpr size_t kScanMin = (size_t)1 << 17;   // doubles (1 MB): below this an array is simply uploaded

enum class InPolicy {
  Plain,        // upload (or fill when uniform)
  ZeroAbsent,   // as Plain, but an all-(+0.0) array yields nullptr
};

class HostInputs {
 public:
  HostInputs(rrtmg_ctx *ctx, hipStream_t s, const char *prefix, int memspace) : ctx_(ctx), s_(s), prefix_(prefix), memspace_(memspace) {}
  // registers one input; *slot receives the device pointer in finish() (at once for memspace 1 and NULL arrays).
  // value on the device = host value * mul (/ div when div != 0); mul == 0: as given.
  void add(const double **slot, const double *host, size_t n, const char *name, bool required, InPolicy policy = InPolicy::Plain,
           double mul = 0.0, double div = 0.0);
  // scans, uploads, fills; false when something failed (ctx->status / ctx->err say what)
  bool finish();

 private:
  struct Entry {
    const double **slot; const double *host; size_t n; const char *name; InPolicy policy; double mul, div;
    int job = -1;   // index into the scan jobs, or -1: upload without asking
  };
  bool upload(const Entry &e);
  bool fill(const Entry &e, double host_value);
  rrtmg_ctx *ctx_; hipStream_t s_; const char *prefix_; int memspace_;
  std::vector<Entry> entries_;
  bool ok_ = true;
};

// kernels behind it (rrtmg_neighbours.hip)
void launch_fill(hipStream_t s, double *p, size_t n, double value);
void launch_scale(hipStream_t s, double *p, size_t n, double mul, double div);   // p = p * mul (/ div)

}  // namespace rrtmg
