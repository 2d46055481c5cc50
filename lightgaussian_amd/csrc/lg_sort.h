// lg_sort.h -- K4 driver: rocPRIM's onesweep radix-sort KERNELS (histogram + one decoupled-lookback pass per digit) behind
// our own host loop.  rocprim::radix_sort_keys resets its lookback states and its ordered-block-id counter with two
// hipMemsetAsync per pass because it reuses one buffer for all passes: 9 memsets ~ 45 us of a 0.17 ms sort of 4 M keys in
// the kernel trace.  Here every pass has its own lookback states and counter, all cleared by ONE memset up front.
// The kernels, their configuration machinery and the temp-storage types are rocPRIM's (rocprim::detail, ROCm 7.x headers);
// -DLG_SORT_ROCPRIM_HOST falls back to the plain library call.
#pragma once

#include <rocprim/rocprim.hpp>
#include <variant>

struct LgSortLayout {
    size_t hist_off, hist_tmp_off, lookback_off, ids_off, keys_tmp_off, total;
    unsigned num_states;
};
static const unsigned LG_SORT_MAX_PLACES = 8;   // 64 key bits / 8

template <class Config>
static LgSortLayout lg_sort_layout(size_t n, hipStream_t stream, hipError_t& err)
{
    using namespace rocprim::detail;
    using config = wrapped_radix_sort_onesweep_config<Config, uint64_t, rocprim::empty_type>;
    LgSortLayout L{};
    target_arch arch;
    err = host_target_arch(stream, arch);
    if (err != hipSuccess) return L;
    const radix_sort_onesweep_config_params params = dispatch_target_arch<config, false>(arch);
    const unsigned items_per_block = params.sort.block_size * params.sort.items_per_thread;
    const unsigned radix_size = 1u << params.radix_bits_per_place;
    L.num_states = radix_size * (unsigned)((n + items_per_block - 1) / items_per_block);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t r = off; off += (bytes + 255) / 256 * 256; return r; };
    L.hist_off = take((size_t)radix_size * LG_SORT_MAX_PLACES * sizeof(unsigned));
    L.hist_tmp_off = take((size_t)radix_size * sizeof(unsigned));
    L.lookback_off = take((size_t)L.num_states * LG_SORT_MAX_PLACES * sizeof(onesweep_lookback_state));
    L.ids_off = take((size_t)LG_SORT_MAX_PLACES * 64);
    L.keys_tmp_off = take(n * sizeof(uint64_t));
    L.total = off;
    return L;
}

// Stable ascending sort of the bits [begin_bit, end_bit) of n 64-bit keys; keys_in is preserved, result in keys_out.
template <class Config>
static hipError_t lg_onesweep_sort_keys(void* temp, size_t& temp_bytes, uint64_t* keys_in, uint64_t* keys_out, unsigned n,
                                        unsigned begin_bit, unsigned end_bit, hipStream_t stream)
{
    using namespace rocprim::detail;
    using key_type = uint64_t;
    using value_type = rocprim::empty_type;
    using config = wrapped_radix_sort_onesweep_config<Config, key_type, value_type>;
    hipError_t err = hipSuccess;
    const LgSortLayout L = lg_sort_layout<Config>(n ? n : 1, stream, err);
    if (err != hipSuccess) return err;
    if (temp == nullptr) { temp_bytes = L.total; return hipSuccess; }
    if (n == 0) return hipSuccess;
    if (temp_bytes < L.total || n >= (1u << 30)) return hipErrorInvalidValue;

    target_arch arch;
    ROCPRIM_RETURN_ON_ERROR(host_target_arch(stream, arch));
    const radix_sort_onesweep_config_params params = dispatch_target_arch<config, false>(arch);
    const unsigned items_per_block = params.sort.block_size * params.sort.items_per_thread;
    const unsigned radix_size = 1u << params.radix_bits_per_place;
    const unsigned places = (end_bit - begin_bit + params.radix_bits_per_place - 1) / params.radix_bits_per_place;
    if (places == 0 || places > LG_SORT_MAX_PLACES) return hipErrorInvalidValue;

    char* base = (char*)temp;
    unsigned* hist = (unsigned*)(base + L.hist_off);
    unsigned* hist_tmp = (unsigned*)(base + L.hist_tmp_off);
    onesweep_lookback_state* lookback = (onesweep_lookback_state*)(base + L.lookback_off);
    char* ids = base + L.ids_off;
    key_type* keys_tmp = (key_type*)(base + L.keys_tmp_off);
    value_type* no_values = nullptr;

    // one clear for the lookback states and block-id counters of ALL passes (contiguous by construction)
    ROCPRIM_RETURN_ON_ERROR(hipMemsetAsync(lookback, 0, (L.ids_off - L.lookback_off) + (size_t)LG_SORT_MAX_PLACES * 64, stream));
    rocprim::identity_decomposer decomposer;
    ROCPRIM_RETURN_ON_ERROR((radix_sort_onesweep_global_offsets<Config, false>(keys_in, no_values, hist, (unsigned)n, places, decomposer,
                                                                               begin_bit, end_bit, stream, false)));
    bool use_atomic = false;
    ROCPRIM_RETURN_ON_ERROR(check_if_using_atomic_block_id(stream, use_atomic));
    const auto variant = constexpr_value_variant<bool, false, true>::create(use_atomic);
    const unsigned blocks = (n + items_per_block - 1) / items_per_block;
    const unsigned full_blocks = n % items_per_block == 0 ? blocks : blocks - 1;

    return std::visit(
        [&](auto use_atomic_block_id) -> hipError_t {
            using ordered_bid_type = block_id_wrapper<unsigned int, use_atomic_block_id>;
            bool to_output = (places - 1) % 2 == 0;      // ping-pong between keys_tmp and keys_out so that the last pass lands in keys_out
            bool from_input = true;
            unsigned bit = begin_bit;
            for (unsigned place = 0; place < places; place++, bit += params.radix_bits_per_place) {
                const unsigned current_radix_bits = std::min(params.radix_bits_per_place, end_bit - bit);
                auto ordered_bid = ordered_bid_type::create(ids + (size_t)place * 64);
                onesweep_lookback_state* states = lookback + (size_t)place * L.num_states;
                unsigned* offsets_in = hist + (size_t)place * radix_size;
                const key_type* src = from_input ? keys_in : (to_output ? keys_tmp : keys_out);
                key_type* dst = to_output ? keys_out : keys_tmp;
                auto kernel = [=](auto arch_config) {
                    static constexpr auto p = decltype(arch_config)::params;
                    onesweep_iteration<p.sort.block_size, p.sort.items_per_thread, p.radix_bits_per_place, false, p.radix_rank_algorithm>(
                        src, dst, no_values, no_values, n, offsets_in, hist_tmp, states, decomposer, bit, current_radix_bits, full_blocks,
                        ordered_bid);
                };
                ROCPRIM_RETURN_ON_ERROR((execute_launch_plan<config, decltype(kernel), radix_sort_onesweep_sort_config_selector>(
                    arch, kernel, dim3(blocks), dim3(params.sort.block_size), 0, stream)));
                from_input = false;
                to_output = !to_output;
            }
            return hipSuccess;
        },
        variant);
}
