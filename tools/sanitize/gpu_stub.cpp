// stand-ins for GpuTiler's device passes: the sanitizer build only runs the host builders (gpu == nullptr everywhere)
#include "stream_tiles.h"
#include "tiles_common.h"
#include "gpu_tiles.h"
namespace hisparse { namespace dev {
GpuTiler::~GpuTiler() {}
GpuTiler::GpuTiler(const detail::Layout&, const void* const[NUM_HBM_CHANNELS], const uint64_t[NUM_HBM_CHANNELS], hipStream_t) {}
GpuTiler::GpuTiler(const detail::Layout&, const CsrView&, hipStream_t) {}
bool GpuTiler::count_rows(std::vector<uint32_t>&, uint64_t&) { return false; }
bool GpuTiler::count_tiles(const std::vector<uint32_t>&, uint32_t, std::vector<uint32_t>&) { return false; }
bool GpuTiler::sort_elements(const std::vector<uint32_t>&, const std::vector<uint32_t>&, const std::vector<uint32_t>&, const std::vector<detail::UnitPlan>&, bool&) { return false; }
bool GpuTiler::delta_slots(std::vector<detail::UnitPlan>&) { return false; }
bool GpuTiler::owner_shares(std::vector<detail::UnitPlan>&, uint32_t) { return false; }
bool GpuTiler::emit(StreamFormat, uint64_t, uint64_t, const std::vector<detail::UnitPlan>&, const std::vector<uint32_t>&, const std::vector<Block>&, bool) { return false; }
bool GpuTiler::bitmap_slice_counts(uint32_t, uint32_t, std::vector<uint32_t>&) { return false; }
bool GpuTiler::bitmap_emit(uint32_t, uint32_t, const std::vector<uint32_t>&, const std::vector<BitmapBlock>&, const std::vector<uint64_t>&, uint64_t, uint64_t, const std::vector<BitmapRun>&, std::vector<uint32_t>&, std::vector<uint64_t>&, const MfmaImage*, bool&) { return false; }
bool GpuTiler::sweep_line_counts(uint32_t, std::vector<uint64_t>&) { return false; }
bool GpuTiler::sweep_sort(const std::vector<uint32_t>&, const std::vector<uint32_t>&, const std::vector<uint32_t>&, uint32_t, std::vector<uint64_t>&, bool&) { return false; }
bool GpuTiler::sweep_emit(const std::vector<SweepBlock>&, uint64_t, uint64_t) { return false; }
}}
