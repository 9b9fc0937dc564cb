// op_join.cpp -- HashBuilderOperator + LookupJoinOperator (inner equi-join) on device.
//
// Reference path replaced:
//   HashBuilderOperator.addInput / finish -> PagesIndex.addPage / createLookupSourceSupplier
//     (…/operator/join/HashBuilderOperator.java:332-364, 492-510; …/operator/PagesIndex.java:212, 492-536)
//   PagesHash ctor / getAddressIndex, ArrayPositionLinks (…/operator/join/PagesHash.java:54-126, 158-170;
//     …/operator/join/ArrayPositionLinks.java:38-50, 90-100)
//   LookupJoinOperator / DefaultPageJoiner.processProbe / JoinProbe / LookupJoinPageBuilder
//     (…/operator/join/LookupJoinOperator.java:46-125; DefaultPageJoiner.java:236-320; JoinProbe.java:56-117;
//      LookupJoinPageBuilder.java:76-139)
// Output rows come out exactly in the reference's order: ascending probe position, and for one probe row its
// matches in descending build position (the chain order).  The whole probe page is joined at once; the
// reference's 1 MB / 8192-row output flushes only move page boundaries.
#include <atomic>
#include <mutex>

#include "comm.hpp"
#include "dynfilter_kernels.hpp"
#include "exprgen.hpp"
#include "exchange_kernels.hpp"
#include "join_kernels.hpp"
#include "join_source.hpp"
#include "operator.hpp"
#include "scan_kernels.hpp"
#include "static_kernels.hpp"

namespace pa {
namespace {

// InterpretedHashGenerator over the join channels, or the precomputed $hashvalue channel
void fill_raw_hash(LookupSourceImpl& ls, const JoinKeys& bk, int32_t n, hipStream_t s)
{
    ls.raw_hash.ensure((size_t)std::max(n, 1) * 8);
    if (n <= 0) return;
    if (ls.hash_channel >= 0) {
        PA_HIP(hipMemcpyAsync(ls.raw_hash.ptr(), ls.cols[ls.hash_channel].values.ptr(), (size_t)n * 8, hipMemcpyDeviceToDevice, s));
        return;
    }
    HashPageArgs ha;
    memset(&ha, 0, sizeof ha);
    for (int i = 0; i < bk.ncols; i++) {
        ha.col[i].values = bk.col[i].values;
        ha.col[i].offsets = bk.col[i].offsets;
        ha.col[i].nulls = bk.col[i].nulls;
        ha.col[i].type = bk.col[i].type;
    }
    ha.ncols = bk.ncols;
    ha.n = n;
    ha.out = ls.raw_hash.as<int64_t>();
    launch_hash_page(ha, s);
}

// fastutil HashCommon.arraySize(expected, 0.75f) as used by PagesHash.java:64
uint32_t array_size(int64_t expected)
{
    uint64_t need = (uint64_t)((expected + 0.75 - 1e-9) / 0.75);  // ceil(expected / 0.75)
    while ((double)need * 0.75 < (double)expected) need++;
    uint64_t s = 2;
    while (s < need) s <<= 1;
    PA_REQUIRE(s <= (1ULL << 30), PA_ERR_INSUFFICIENT_RESOURCES, "Size of hash table cannot exceed 1 billion entries");
    return (uint32_t)s;
}

class HashBuilderOperator : public pa_operator {
public:
    hipStream_t main_stream() override { return stream_.get(); }
    HashBuilderOperator(const pa_hash_builder_desc* d, pa_lookup_source* bridge) : stream_(d->stream)
    {
        PA_REQUIRE(d != nullptr, PA_ERR_INVALID_ARGUMENT, "descriptor is null");
        require_device();
        PA_REQUIRE(d->input_channel_count > 0 && d->input_channel_count <= 32, PA_ERR_NOT_SUPPORTED, "1..32 build channels");
        PA_REQUIRE(d->join_channel_count > 0 && d->join_channel_count <= kMaxJoinChannels, PA_ERR_NOT_SUPPORTED, "1..8 join channels");
        ls_ = std::make_shared<LookupSourceImpl>();
        ls_->cols.resize(d->input_channel_count);
        for (int c = 0; c < d->input_channel_count; c++) {
            ls_->cols[c].type = d->input_types[c];
            ls_->cols[c].varwidth = d->input_types[c] == PA_VARCHAR;
        }
        for (int i = 0; i < d->join_channel_count; i++) {
            PA_REQUIRE(d->join_channels[i] >= 0 && d->join_channels[i] < d->input_channel_count, PA_ERR_INVALID_ARGUMENT, "join channel out of range");
            ls_->join_channels.push_back(d->join_channels[i]);
        }
        for (int i = 0; i < d->output_channel_count; i++) {
            PA_REQUIRE(d->output_channels[i] >= 0 && d->output_channels[i] < d->input_channel_count, PA_ERR_INVALID_ARGUMENT, "output channel out of range");
            ls_->output_channels.push_back(d->output_channels[i]);
        }
        ls_->hash_channel = d->hash_channel;
        PA_REQUIRE(ls_->hash_channel < d->input_channel_count, PA_ERR_INVALID_ARGUMENT, "hash channel out of range");
        PA_REQUIRE(ls_->hash_channel < 0 || d->input_types[ls_->hash_channel] == PA_BIGINT, PA_ERR_INVALID_ARGUMENT, "hash channel must be BIGINT");
        expected_ = d->expected_positions;
        bridge->impl = ls_;
        ctl_ = static_cast<int32_t*>(ctl_buf_.ensure(64));
        PA_HIP(hipMemsetAsync(ctl_, 0, 64, stream_.get()));
    }
    ~HashBuilderOperator() override { (void)hipStreamSynchronize(stream_.get()); }

    bool needs_input() override { return !finishing_; }
    bool takes_retained() override { return true; }

    // PagesIndex.addPage: append every channel to the flat build columns
    void add_input(const pa_page* page) override
    {
        PA_REQUIRE(!finishing_, PA_ERR_ILLEGAL_STATE, "Operator is already finishing");
        PA_REQUIRE(page != nullptr, PA_ERR_INVALID_ARGUMENT, "page is null");
        PA_REQUIRE(page->channel_count == (int32_t)ls_->cols.size(), PA_ERR_INVALID_ARGUMENT, "page channel count does not match the build types");
        const int64_t m = page->position_count;
        const bool retained = (page->flags & PA_PAGE_RETAINED) != 0 && page->release != nullptr;
        if (m == 0) {
            if (retained) page->release(page->release_ctx);
            return;
        }
        // The first page of a build side, kept alive by its owner (PA_PAGE_RETAINED): PagesIndex would hold on to the Page's blocks --
        // so do the build columns: the flat block arrays are read in place until the lookup source is destroyed (a second page
        // moves them into allocations of their own, reserve_keep).  A build side that arrives as one page is never copied.
        if (retained && ls_->n == 0 && page->mem == PA_MEM_DEVICE && borrow_page(page)) return;
        // (a retained page that is copied after all goes back to its owner when the copies have run -- also when a check below throws)
        struct ReleaseAtExit {
            const pa_page* p;
            bool on;
            hipStream_t s;
            ~ReleaseAtExit()
            {
                if (!on) return;
                (void)hipStreamSynchronize(s);
                p->release(p->release_ctx);
            }
        } release_at_exit{page, retained, stream_.get()};
        PA_REQUIRE((int64_t)ls_->n + m <= INT32_MAX, PA_ERR_INSUFFICIENT_RESOURCES, "build side exceeds 2^31 positions");
        hipStream_t s = stream_.get();
        DevPage dp = stager_.stage(page, nullptr, s);
        const int64_t n0 = ls_->n;
        const int64_t want = std::max<int64_t>(n0 + m, expected_);
        for (size_t c = 0; c < ls_->cols.size(); c++) {
            BuildColumn& bc = ls_->cols[c];
            const DevColumn& in = dp.cols[c];
            PA_REQUIRE(in.type == bc.type, PA_ERR_INVALID_ARGUMENT, "page block type does not match the declared build type");
            if (bc.varwidth) {
                int32_t ends[2] = {0, 0};
                PA_HIP(hipMemcpyAsync(&ends[0], in.offsets, 4, hipMemcpyDeviceToHost, s));
                PA_HIP(hipMemcpyAsync(&ends[1], in.offsets + m, 4, hipMemcpyDeviceToHost, s));
                PA_HIP(hipStreamSynchronize(s));
                int64_t add = (int64_t)ends[1] - ends[0];
                PA_REQUIRE(bc.bytes + add <= INT32_MAX, PA_ERR_INSUFFICIENT_RESOURCES, "build VARCHAR column exceeds 2 GB");
                bc.offsets.reserve_keep((size_t)(want + 1) * 4, (size_t)(n0 + 1) * 4, s);
                bc.values.reserve_keep((size_t)std::max<int64_t>(bc.bytes + add, 16), (size_t)bc.bytes, s);
                launch_rebase_offsets(in.offsets, ends[0], (int32_t)bc.bytes, m + 1, bc.offsets.as<int32_t>() + n0, s);
                if (add) PA_HIP(hipMemcpyAsync(bc.values.as<char>() + bc.bytes, static_cast<const char*>(in.values) + ends[0], (size_t)add, hipMemcpyDeviceToDevice, s));
                bc.bytes += add;
            }
            else {
                int w = type_width(bc.type);
                bc.values.reserve_keep((size_t)want * w, (size_t)n0 * w, s);
                PA_HIP(hipMemcpyAsync(bc.values.as<char>() + n0 * w, in.values, (size_t)m * w, hipMemcpyDeviceToDevice, s));
            }
            if (in.nulls || bc.has_nulls) {
                bc.nulls.reserve_keep((size_t)want, bc.has_nulls ? (size_t)n0 : 0, s);
                if (!bc.has_nulls && n0) PA_HIP(hipMemsetAsync(bc.nulls.ptr(), 0, (size_t)n0, s));
                if (in.nulls) PA_HIP(hipMemcpyAsync(bc.nulls.as<char>() + n0, in.nulls, (size_t)m, hipMemcpyDeviceToDevice, s));
                else PA_HIP(hipMemsetAsync(bc.nulls.as<char>() + n0, 0, (size_t)m, s));
                bc.has_nulls = true;
            }
        }
        ls_->n = (int32_t)(n0 + m);
        // staged host pages live in the stager's arena, which the next add_input overwrites.  (A device page is its producer's again
        // with that operator's next call: on the caller's stream these copies are ordered before it, a stream of this operator's own
        // is drained by pa_op_add_input -- abi.cpp.)  A retained page that was copied is handed back once the copies have run.
        if (page->mem != PA_MEM_DEVICE) PA_HIP(hipStreamSynchronize(s));
    }

    // true: every channel of the page is a flat or variable-width block the build columns can read where it is
    bool borrow_page(const pa_page* page)
    {
        const int64_t m = page->position_count;
        for (size_t c = 0; c < ls_->cols.size(); c++) {
            const pa_column& in = page->columns[c];
            const BuildColumn& bc = ls_->cols[c];
            if (in.type != bc.type || in.values == nullptr) return false;
            if (bc.varwidth ? in.encoding != PA_VARWIDTH : in.encoding != PA_FLAT) return false;
            if (bc.varwidth) return false;   // (a block's offsets need not start at 0: variable-width channels are appended as before)
        }
        for (size_t c = 0; c < ls_->cols.size(); c++) {
            const pa_column& in = page->columns[c];
            BuildColumn& bc = ls_->cols[c];
            bc.values.borrow(in.values, (size_t)m * type_width(bc.type));
            if (in.nulls) {
                bc.nulls.borrow(in.nulls, (size_t)m);
                bc.has_nulls = true;
            }
        }
        ls_->n = (int32_t)m;
        ls_->releases.push_back(LookupSourceImpl::Release{page->release, page->release_ctx});
        return true;
    }

    // finishInput -> buildLookupSource -> new PagesHash(...)
    void finish() override
    {
        if (finishing_) return;
        finishing_ = true;
        hipStream_t s = stream_.get();
        const int32_t n = ls_->n;
        const uint32_t hash_size = array_size(n);
        ls_->mask = hash_size - 1;
        ls_->links.ensure((size_t)std::max(n, 1) * 4);
        ls_->slot_of.ensure((size_t)std::max(n, 1) * 4);
        JoinKeys bk = ls_->build_keys();
        ls_->keyed = bk.ncols == 1 && (bk.col[0].type == PA_BIGINT || bk.col[0].type == PA_INTEGER || bk.col[0].type == PA_DATE);
        // the probe-side table: its own size (load <= 1/2), whatever PagesHash.key[] -- the reference's layout -- uses
        uint64_t slots = 1024;
        while (slots < 2 * (uint64_t)std::max(n, 1)) slots <<= 1;
        PA_REQUIRE(slots <= (1ULL << 31), PA_ERR_INSUFFICIENT_RESOURCES, "join build side too large");
        ls_->probe_mask = (uint32_t)(slots - 1);
        bool keyed_dups = false, rank_pending = false;
        auto build_table = [&] {
            const int32_t dups = build_key_slots(bk.col[0], n, slots, s);
            // chains exist only when some key has several rows: two passes over the rows and the table that unique keys -- the build
            // side of a primary-key join -- do without
            if (dups > 0 && !links_built_) launch_join_keyed_links(n, ls_->key_slots.as<JoinKeySlot>(), ls_->probe_mask, ls_->slot_of.as<int32_t>(), ls_->links.as<int32_t>(), s);
            keyed_dups = dups > 0;   // (-1: a partitioned build is in flight, settle_partitioned_build reads its verdict)
        };
        // The partitioned build leaves its verdict in the control words the read-back below fetches anyway -- [1] some key has
        // several rows, [2] a partition too full or a chain too long: the table the other way -- so it costs no round trip of its
        // own; its temporaries live until then.
        auto settle_partitioned_build = [&](int32_t* ctl) {
            if (!part_pending_) return;
            part_pending_ = false;
            part_temps_.clear();  // (the stream has drained)
            if (ctl[2] != 0) {
                ls_->probe_wrap = ls_->probe_mask;
                const int32_t dups = build_key_slots_plain(bk.col[0], n, ls_->key_slots.as<JoinKeySlot>(), s);
                if (dups) launch_join_keyed_links(n, ls_->key_slots.as<JoinKeySlot>(), ls_->probe_mask, ls_->slot_of.as<int32_t>(), ls_->links.as<int32_t>(), s);
                keyed_dups = dups != 0;
                int32_t again[1] = {0};
                read_back(again, ctl_, 4, s);
                ctl[0] = again[0];
            }
            else keyed_dups = ctl[1] != 0;
        };
        timer.begin(s);
        if (ls_->keyed) {
            // one integer key.  Key range and bitmap first: over a dense enough key range without NULL or duplicate keys the key rank
            // index is the whole lookup structure and no table is built (whether it holds is read back with the error word below)
            if (n > 0) build_key_bitmap(bk.col[0], n, s);
            else PA_HIP(hipStreamSynchronize(s));
            rank_pending = start_rank_index(bk.col[0], n, s);
            if (!rank_pending) build_table();
        }
        else {
            compute_raw_hash(bk, n, s);
            ls_->key.ensure((size_t)hash_size * 4);
            launch_join_build(bk, ls_->raw_hash.as<int64_t>(), n, ls_->key.as<int32_t>(), ls_->mask, ls_->slot_of.as<int32_t>(), ls_->links.as<int32_t>(),
                              ctl_, s);
            ls_->reference_built = true;
            launch_join_tag_slots(ls_->key.as<int32_t>(), (int64_t)hash_size, ls_->raw_hash.as<int64_t>(),
                                  static_cast<uint64_t*>(ls_->tagged.ensure((size_t)slots * 8)), ls_->probe_mask, s);
        }
        timer.end(s);
        int32_t ctl[6] = {0, 0, 0, 0, 0, 0};  // [0] error word, [4] distinct build keys, [5] some build row not at its key's rank
        read_back(ctl, ctl_, sizeof ctl, s);
        settle_partitioned_build(ctl);
        if (rank_pending && !finish_rank_index(n, ctl[4], ctl[5] != 0)) {
            timer.begin(s);
            build_table();
            timer.end(s);
            read_back(ctl, ctl_, 12, s);
            settle_partitioned_build(ctl);
        }
        const int32_t err = ctl[0];
        ls_->has_duplicates = !ls_->keyed || keyed_dups;
        ls_->error.store(err);
        ls_->built.store(true);  // lendPartitionLookupSource: probes may proceed
        if (err) throw Error(err, "hash build failed on device");
    }

    // The probe-side table straight from the build rows (the raw hash from the $hashvalue channel when there is one, else from the
    // key inside the kernel); PagesHash.key[] is built when somebody asks for it.  Returns whether some key has several rows.
    int32_t build_key_slots(const JoinCol& key, int32_t n, uint64_t slots, hipStream_t s)
    {
        const int64_t* raw = nullptr;
        if (ls_->hash_channel >= 0 && n > 0) raw = ls_->cols[ls_->hash_channel].values.as<int64_t>();
        JoinKeySlot* table = static_cast<JoinKeySlot*>(ls_->key_slots.ensure((size_t)slots * sizeof(JoinKeySlot)));
        ls_->probe_wrap = ls_->probe_mask;
        // large build sides: rows regrouped by the partition of their home slot, tables assembled in LDS and written once
        // (join_kernels.hip).  It leaves no slot_of, so chains (duplicate keys) -- and partitions too full, which a decent hash
        // does not produce -- are built the other way
        const int64_t partitions = (int64_t)slots >> kJoinPartSlotsLog2;
        // (not with a $hashvalue channel: the partitioned build places rows by the hash it computes from the key itself, the
        // probe by the channel's value -- the two must be the same function)
        if (raw == nullptr && n >= (1 << 20) && partitions >= 2 && partitions <= 4096 && !getenv("PRESTO_AMD_NO_PARTITIONED_BUILD")) {
            partitioned_build(key, n, table, (int32_t)partitions, s);
            ls_->probe_wrap = (uint32_t)kJoinPartSlots - 1u;
            links_built_ = true;  // (chains of keys with several rows included)
            return -1;            // the verdict travels with the caller's read-back of the control words
        }
        return build_key_slots_plain(key, n, table, s);
    }
    // slot by slot, straight from the build rows; returns whether some key has several rows (one round trip)
    int32_t build_key_slots_plain(const JoinCol& key, int32_t n, JoinKeySlot* table, hipStream_t s)
    {
        const int64_t* raw = nullptr;
        if (ls_->hash_channel >= 0 && n > 0) raw = ls_->cols[ls_->hash_channel].values.as<int64_t>();
        links_built_ = false;
        int32_t dups = 0;
        PA_HIP(hipMemsetAsync(ctl_ + 1, 0, 8, s));
        launch_join_keyed_build(key, raw, n, table, ls_->probe_mask, ls_->slot_of.as<int32_t>(), ls_->links.as<int32_t>(), ctl_, s);
        read_back(&dups, ctl_ + 1, 4, s);
        return dups;
    }

    // The key rank index (join_kernels.hpp) from the key bitmap: enqueues its construction; false = the build side does not qualify up
    // front (no bitmap, NULL keys).  Whether it holds -- no key on several rows -- shows in ctl_[4] (distinct keys) once the stream has
    // drained: finish_rank_index.  PRESTO_AMD_NO_RANK_INDEX=1 turns it off (A/B runs, tests of the table builds).
    bool start_rank_index(const JoinCol& key, int32_t n, hipStream_t s)
    {
        if (ls_->bitmap.bits == nullptr || key.nulls != nullptr || n <= 0 || duplicates_certain_ || getenv("PRESTO_AMD_NO_RANK_INDEX")) {
            pair_keys_.release();
            pair_rows_.release();
            pair_first_.release();
            return false;
        }
        const int64_t nwords = (int64_t)(ls_->bitmap.range >> 6) + 1;
        JoinRankWord* words = static_cast<JoinRankWord*>(ls_->rank_words.ensure((size_t)nwords * sizeof(JoinRankWord)));
        PA_HIP(hipMemsetAsync(ctl_ + 4, 0, 8, s));
        const int64_t tiles = join_rank_tiles(nwords);
        launch_join_rank_words(ls_->bitmap.bits, nwords, words, static_cast<int32_t*>(rank_counts_.ensure((size_t)tiles * 4)),
                               rank_temp_.ensure(scan_temp_bytes(tiles)), ctl_ + 4, s);
        // (rows of duplicate keys overwrite each other here: the index is dropped then)
        if (pairs_ == n) {  // the pairs regrouped by key range: the scatter stays inside one partition's slice at a time
            launch_join_rank_rows_pairs(pair_keys_.as<uint64_t>(), pair_rows_.as<int32_t>(), n, words, ls_->bitmap.min_key,
                                        static_cast<int32_t*>(ls_->rank_rows.ensure((size_t)n * 4)), s, pair_first_.as<int64_t>(), pair_partitions_, pair_shift_, ctl_ + 4);
            PA_HIP(hipMemsetAsync(ctl_ + 5, 1, 4, s));   // "out of key order" (non-zero), by the check that brought us here
        }
        else if (keys_ascending_) {
            // no key is smaller than its predecessor (k_join_key_disorder): with no key on two rows -- the condition the index stands
            // on anyway -- the keys are strictly ascending and the rank of a key IS its row; nothing to write, nothing to check
            PA_HIP(hipMemsetAsync(ctl_ + 5, 0, 4, s));
        }
        else launch_join_rank_rows(key, n, words, ls_->bitmap.min_key, static_cast<int32_t*>(ls_->rank_rows.ensure((size_t)n * 4)), ctl_ + 5, s, ctl_ + 4);
        pair_keys_.release();
        pair_rows_.release();
        pair_first_.release();
        launch_fill_i32(ls_->links.as<int32_t>(), -1, n, s);
        return true;
    }
    bool finish_rank_index(int32_t n, int32_t distinct, bool unordered)
    {
        rank_counts_.release();
        rank_temp_.release();
        if (distinct != n) {  // some key on several rows
            ls_->rank_words.release();
            ls_->rank_rows.release();
            return false;
        }
        if (!unordered) ls_->rank_rows.release();  // build rows in key order: rank == build position
        ls_->rank = JoinRankIndex{ls_->rank_words.as<JoinRankWord>(), unordered ? ls_->rank_rows.as<int32_t>() : nullptr, ls_->bitmap.min_key, ls_->bitmap.range};
        return true;
    }

    // Enqueues the keyed probe-side table built partition by partition.  ctl_[1] != 0: some key has several rows -- their chains
    // (`links`, the slots' `next`) were built in the same pass (otherwise links stay -1); ctl_[2] != 0: not built (see the kernel)
    void partitioned_build(const JoinCol& key, int32_t n, JoinKeySlot* table, int32_t partitions, hipStream_t s)
    {
        part_temps_.clear();
        part_temps_.resize(8);
        DevBuf &part = part_temps_[0], &keys_in = part_temps_[1], &keys_out = part_temps_[2], &rows_in = part_temps_[3], &rows_out = part_temps_[4],
               &counts = part_temps_[5], &first = part_temps_[6], &temp = part_temps_[7];
        int32_t* pid = static_cast<int32_t*>(part.ensure((size_t)n * 4));
        uint64_t* kin = static_cast<uint64_t*>(keys_in.ensure((size_t)n * 8));
        int32_t* rin = static_cast<int32_t*>(rows_in.ensure((size_t)n * 4));
        launch_join_part_ids(key, n, ls_->probe_mask, pid, kin, rin, s);
        MsplitCol cols[2];
        memset(cols, 0, sizeof cols);
        cols[0].in = kin;
        cols[0].out = keys_out.ensure((size_t)n * 8);
        cols[0].width = 8;
        cols[1].in = rin;
        cols[1].out = rows_out.ensure((size_t)n * 4);
        cols[1].width = 4;
        int64_t* cnt = static_cast<int64_t*>(counts.ensure((size_t)(partitions + 1) * 8));
        PA_HIP(hipMemsetAsync(cnt, 0, (size_t)(partitions + 1) * 8, s));
        launch_msplit(pid, n, partitions, cols, 2, cnt, temp.ensure(msplit_temp_bytes(n, partitions)), s);
        int64_t* fst = static_cast<int64_t*>(first.ensure((size_t)(partitions + 1) * 8));
        launch_exclusive_prefix_i64(cnt, partitions, fst, s);  // fst[p] = first row of partition p, fst[partitions] = n
        PA_HIP(hipMemsetAsync(ctl_ + 1, 0, 8, s));
        launch_fill_i32(ls_->links.as<int32_t>(), -1, n, s);
        launch_join_part_build(keys_out.as<uint64_t>(), rows_out.as<int32_t>(), fst, partitions, ls_->probe_mask, table, ls_->links.as<int32_t>(), ctl_, s);
        part_pending_ = true;  // ctl_[1]: some key has several rows; ctl_[2]: start over the other way (finish() reads both)
    }

    void compute_raw_hash(const JoinKeys& bk, int32_t n, hipStream_t s) { fill_raw_hash(*ls_, bk, n, s); }

    // min / max of the build keys (one reduction pass), then -- when the keys are dense enough for the bitmap to be smaller than
    // the slot table -- one bit per existing key
    void build_key_bitmap(const JoinCol& key, int32_t n, hipStream_t s)
    {
        DevBuf running, partials;
        uint64_t* run = static_cast<uint64_t*>(running.ensure(64));
        launch_join_key_stats(key, n, run, partials.ensure(join_key_stats_temp_bytes()), s);
        uint64_t raw[4];
        read_back(raw, run, 32, s);
        const JoinKeyStats st = join_key_stats_decode(raw);
        if (!st.any) return;  // every key NULL
        keys_ascending_ = !st.descending;
        ls_->key_range_valid = true;
        ls_->key_min = st.min_key;
        ls_->key_max = st.max_key;
        const int64_t h[4] = {st.min_key, st.max_key, 1, st.descending ? 1 : 0};
        const uint64_t range = (uint64_t)h[1] - (uint64_t)h[0];
        if (range >= 64ULL * (uint64_t)n || range >= (1ULL << 36)) return;
        uint64_t* bits = static_cast<uint64_t*>(ls_->key_bits.ensure((size_t)((range >> 6) + 1) * 8));
        const int shift = join_range_shift(range);
        // more rows than key values: some key is on several rows for certain, so there will be no rank index and no use for the pairs
        // regrouped by key range -- the bitmap is small against the rows (its words stay in the L2) and takes the rows' atomics as they
        // come when they mostly ascend (8 M rows over 1.6 M keys: 0.33 ms of range passes -> 0.04 ms)
        duplicates_certain_ = range + 1 < (uint64_t)n;
        // (rows that mostly ascend -- fewer than one descent per four waves: sequences of pages that wrap around -- keep the atomics few:
        // neighbouring lanes set bits of the same word and combine them first; random rows go through the range passes all the same)
        const bool atomics_do = duplicates_certain_ && st.descents * 256 < (uint64_t)n;
        if (h[3] != 0 && n >= (1 << 18) && shift >= 0 && key.nulls == nullptr && !atomics_do && !getenv("PRESTO_AMD_NO_RANGE_BITMAP")) {
            // rows out of key order: regroup the (key, row) pairs by key range, OR the bits in LDS (join_kernels.hip); the pairs stay
            // for the rank -> row array of the key rank index
            const int32_t partitions = (int32_t)((range >> shift) + 1);
            DevBuf part, keys_in, rows_in, counts, temp;
            int32_t* pid = static_cast<int32_t*>(part.ensure((size_t)n * 4));
            uint64_t* kin = static_cast<uint64_t*>(keys_in.ensure((size_t)n * 8));
            int32_t* rin = static_cast<int32_t*>(rows_in.ensure((size_t)n * 4));
            launch_join_range_ids(key, n, h[0], shift, partitions, pid, kin, rin, s);
            MsplitCol cols[2];
            memset(cols, 0, sizeof cols);
            cols[0].in = kin;
            cols[0].out = pair_keys_.ensure((size_t)n * 8);
            cols[0].width = 8;
            cols[1].in = rin;
            cols[1].out = pair_rows_.ensure((size_t)n * 4);
            cols[1].width = 4;
            int64_t* cnt = static_cast<int64_t*>(counts.ensure((size_t)(partitions + 2) * 8));
            launch_msplit(pid, n, partitions + 1, cols, 2, cnt, temp.ensure(msplit_temp_bytes(n, partitions + 1)), s);
            int64_t* fst = static_cast<int64_t*>(pair_first_.ensure((size_t)(partitions + 3) * 8));
            launch_exclusive_prefix_i64(cnt, partitions + 1, fst, s);
            launch_join_range_bitmap(pair_keys_.as<uint64_t>(), fst, partitions, h[0], shift, range, bits, s);
            pairs_ = n;
            pair_partitions_ = partitions;
            pair_shift_ = shift;
        }
        else {
            launch_join_key_bitmap(key, n, h[0], range, bits, s);
        }
        ls_->bitmap = JoinKeyBitmap{bits, h[0], range};
    }

    bool get_output(pa_page*) override { return false; }
    bool is_finished() override { return finishing_; }
    int64_t memory_bytes() override
    {
        int64_t b = (int64_t)(ls_->key.capacity() + ls_->links.capacity() + ls_->raw_hash.capacity() + ls_->slot_of.capacity() + ls_->tagged.capacity() + ls_->key_slots.capacity() + ls_->key_bits.capacity() + ls_->rank_words.capacity() + ls_->rank_rows.capacity());
        for (const auto& c : ls_->cols) b += (int64_t)(c.values.capacity() + c.offsets.capacity() + c.nulls.capacity());
        return b;
    }

private:
    Stream stream_;
    PageStager stager_;
    std::shared_ptr<LookupSourceImpl> ls_;
    DevBuf ctl_buf_, rank_counts_, rank_temp_;
    // (key, row) pairs regrouped by key range, kept between build_key_bitmap and start_rank_index (rows out of key order)
    DevBuf pair_keys_, pair_rows_, pair_first_;
    bool keys_ascending_ = false;  // no build key is smaller than the key of the row before it (build_key_bitmap)
    bool duplicates_certain_ = false;  // fewer key values between min and max than rows (build_key_bitmap)
    bool links_built_ = false;  // the table build left the chains of keys with several rows behind (the partitioned build does)
    bool part_pending_ = false; // a partitioned build is in flight: its verdict is in ctl_[1..2]
    std::vector<DevBuf> part_temps_;  // its temporaries, released once the stream has drained
    int32_t pairs_ = 0, pair_partitions_ = 0;
    int pair_shift_ = 0;
    int32_t* ctl_ = nullptr;
    int64_t expected_ = 0;
    bool finishing_ = false;
};

class LookupJoinOperator : public pa_operator {
public:
    LookupJoinOperator(const pa_lookup_join_desc* d, pa_lookup_source* bridge) : stream_(d->stream)
    {
        PA_REQUIRE(d != nullptr, PA_ERR_INVALID_ARGUMENT, "descriptor is null");
        require_device();
        PA_REQUIRE(bridge->impl != nullptr, PA_ERR_ILLEGAL_STATE, "lookup source has no build operator yet");
        ls_ = bridge->impl;
        PA_REQUIRE(d->join_channel_count == (int32_t)ls_->join_channels.size(), PA_ERR_INVALID_ARGUMENT, "probe and build join channel counts differ");
        n_probe_channels_ = d->probe_channel_count;
        probe_types_.assign(d->probe_types, d->probe_types + d->probe_channel_count);
        for (int i = 0; i < d->join_channel_count; i++) {
            int c = d->probe_join_channels[i];
            PA_REQUIRE(c >= 0 && c < n_probe_channels_, PA_ERR_INVALID_ARGUMENT, "probe join channel out of range");
            PA_REQUIRE(probe_types_[c] == ls_->cols[ls_->join_channels[i]].type, PA_ERR_INVALID_ARGUMENT, "probe / build join key types differ");
            join_channels_.push_back(c);
        }
        hash_channel_ = d->probe_hash_channel;
        for (int i = 0; i < d->probe_output_channel_count; i++) {
            int c = d->probe_output_channels[i];
            PA_REQUIRE(c >= 0 && c < n_probe_channels_, PA_ERR_INVALID_ARGUMENT, "probe output channel out of range");
            output_channels_.push_back(c);
        }
        output_mem_ = d->output_mem;
        PA_REQUIRE(d->join_type >= PA_JOIN_INNER && d->join_type <= PA_JOIN_FULL_OUTER, PA_ERR_INVALID_ARGUMENT, "unknown join type");
        probe_outer_ = d->join_type == PA_JOIN_PROBE_OUTER || d->join_type == PA_JOIN_FULL_OUTER;
        track_visited_ = d->join_type == PA_JOIN_LOOKUP_OUTER || d->join_type == PA_JOIN_FULL_OUTER;
        probe_flags_ = (probe_outer_ ? 1 : 0) | (d->output_single_match ? 2 : 0);
        needed_.assign(n_probe_channels_, false);
        for (int c : join_channels_) needed_[c] = true;
        for (int c : output_channels_) needed_[c] = true;
        if (d->filter) setup_filter(*d->filter, d->output_single_match != 0);
        if (hash_channel_ >= 0) needed_[hash_channel_] = true;
        ctl_ = static_cast<int32_t*>(ctl_buf_.ensure(64));
        h_ctl_ = static_cast<int32_t*>(h_ctl_buf_.ensure(64));
        out_cols_.resize(output_channels_.size() + ls_->output_channels.size());
    }
    ~LookupJoinOperator() override { (void)hipStreamSynchronize(stream_.get()); }
    hipStream_t private_stream() override { return stream_.owned() ? stream_.get() : nullptr; }
    hipStream_t main_stream() override { return stream_.get(); }

    // LookupJoinOperator.needsInput: only once the lookup source is ready, one probe page at a time
    bool needs_input() override { return !finishing_ && !pending_ && ls_->built.load(); }
    bool is_blocked() override { return !ls_->built.load(); }

    void add_input(const pa_page* page) override
    {
        PA_REQUIRE(ls_->built.load(), PA_ERR_ILLEGAL_STATE, "lookup source is not built yet");
        PA_REQUIRE(!finishing_ && !pending_, PA_ERR_ILLEGAL_STATE, "Operator does not need input");
        PA_REQUIRE(page != nullptr && page->channel_count == n_probe_channels_, PA_ERR_INVALID_ARGUMENT, "probe page does not match the probe types");
        if (page->position_count == 0) return;
        hipStream_t s = stream_.get();
        in_ = stager_.stage(page, &needed_, s);
        // (a device page some operator upstream returned: its buffers are that operator's again with its next call)
        in_volatile_ = page->mem == PA_MEM_DEVICE && (page->flags & PA_PAGE_STABLE) == 0;
        const int32_t n = in_.n;
        JoinKeys pk;
        memset(&pk, 0, sizeof pk);
        pk.ncols = (int32_t)join_channels_.size();
        for (int i = 0; i < pk.ncols; i++) {
            const DevColumn& c = in_.cols[join_channels_[i]];
            PA_REQUIRE(c.type == probe_types_[join_channels_[i]], PA_ERR_INVALID_ARGUMENT, "page block type does not match the declared probe type");
            pk.col[i].values = c.values;
            pk.col[i].offsets = c.offsets;
            pk.col[i].nulls = c.nulls;
            pk.col[i].type = c.type;
        }
        const int64_t* probe_hash = nullptr;
        if (hash_channel_ >= 0) {
            probe_hash = static_cast<const int64_t*>(in_.cols[hash_channel_].values);
        }
        else if (!ls_->keyed) {
            HashPageArgs ha;
            memset(&ha, 0, sizeof ha);
            for (int i = 0; i < pk.ncols; i++) {
                ha.col[i].values = pk.col[i].values;
                ha.col[i].offsets = pk.col[i].offsets;
                ha.col[i].nulls = pk.col[i].nulls;
                ha.col[i].type = pk.col[i].type;
            }
            ha.ncols = pk.ncols;
            ha.n = n;
            ha.out = static_cast<int64_t*>(hash_.ensure((size_t)n * 8));
            launch_hash_page(ha, s);
            probe_hash = ha.out;
        }
        int32_t* head = static_cast<int32_t*>(head_.ensure((size_t)n * 4));
        int32_t* counts = static_cast<int32_t*>(counts_.ensure((size_t)n * 4));
        timer.begin(s);
        // the matches of the page, in 64 bits: a skewed key or a high fan-out can put more than 2^31 of them behind one page
        if (ls_->keyed) {  // one integer key: key-in-slot table, raw hash computed in the kernel unless a $hashvalue channel came along
            // (the keyed kernel sums the page's matches on the way: one atomic per wave instead of a pass over the counts)
            int64_t* totals = static_cast<int64_t*>(totals_.ensure(16 * 8));
            PA_HIP(hipMemsetAsync(totals, 0, 16 * 8, s));
            int32_t* tile_totals = !ls_->has_duplicates ? static_cast<int32_t*>(tile_totals_.ensure((size_t)join_probe_tiles(n) * 4)) : nullptr;
            tiles_valid_ = launch_join_probe_count_keyed(pk.col[0], probe_hash, n, ls_->key_slots.as<JoinKeySlot>(), ls_->probe_mask, ls_->probe_wrap, ls_->links.as<int32_t>(), ls_->bitmap, ls_->rank, head, counts,
                                                         probe_flags_, s, totals, !ls_->has_duplicates, tile_totals);
            timer.end(s);
            PA_HIP(hipMemcpyAsync(h_totals_.ensure(16 * 8), totals, 16 * 8, hipMemcpyDeviceToHost, s));
            totals_pending_ = true;
            pending_ = true;
            range_lo_ = 0;
            remaining_ = -1;
            eager_ = false;
            if (tiles_valid_ && eager_possible(n)) emit_eagerly(n, s);
            return;
        }
        else {
            launch_join_probe_count(ls_->build_keys(), pk, probe_hash, n, ls_->tagged.as<uint64_t>(), ls_->probe_mask, ls_->links.as<int32_t>(), head, counts,
                                    probe_flags_, s);
            launch_sum_i32_i64(counts, n, reinterpret_cast<int64_t*>(ctl_ + 4), s);
        }
        timer.end(s);
        PA_HIP(hipMemcpyAsync(h_ctl_ + 4, ctl_ + 4, 8, hipMemcpyDeviceToHost, s));
        pending_ = true;
        range_lo_ = 0;
        remaining_ = -1;
    }

    // At most this many output rows per page: the reference's LookupJoinPageBuilder flushes bounded pages (1 MB / 8192 rows);
    // here a probe page whose matches exceed the bound comes out as several pages, each the join of a row range of the page.
    static int64_t max_output_rows()
    {
        const char* e = getenv("PRESTO_AMD_JOIN_MAX_OUTPUT_ROWS");  // tests exercise the splitting with a small bound
        return e ? std::max<int64_t>(strtoll(e, nullptr, 10), 1) : (int64_t)1 << 30;
    }

    bool get_output(pa_page* out) override
    {
        if (!pending_) return false;
        hipStream_t s = stream_.get();
        const int32_t n = in_.n;
        if (eager_) {
            // the pairs and the output columns were enqueued behind the count (emit_eagerly): the page's row count is all that is missing
            PA_HIP(hipStreamSynchronize(s));
            int64_t total = 0;
            for (int i = 0; i < 16; i++) total += h_totals_.as<int64_t>()[i];
            totals_pending_ = false;
            eager_ = false;
            pending_ = false;
            last_matches_ = (int32_t)total;
            if (total == 0) return false;
            publish_output(out_cols_, (int32_t)total, output_mem_, s, out, out_storage_);
            return true;
        }
        if (remaining_ < 0) {
            PA_HIP(hipStreamSynchronize(s));
            if (totals_pending_) {
                remaining_ = 0;
                for (int i = 0; i < 16; i++) remaining_ += h_totals_.as<int64_t>()[i];
                totals_pending_ = false;
            }
            else memcpy(&remaining_, h_ctl_ + 4, 8);
            // more matches than one output page holds: the page is joined range by range over several get_output calls, and the
            // operator upstream may be given its next page in between -- the probe channels still to be read are kept
            if (remaining_ > max_output_rows() && in_volatile_) keep_input(s);
        }
        int32_t lo = range_lo_, hi = n;
        int64_t sum = remaining_;
        for (;;) {
            if (sum == 0 && hi == n) {  // nothing (more) to emit
                pending_ = false;
                last_matches_ = 0;
                return false;
            }
            while (sum > max_output_rows()) {
                PA_REQUIRE(hi - lo > 1, PA_ERR_INSUFFICIENT_RESOURCES, "one probe row has more matches than an output page may hold (2^30)");
                hi = lo + (hi - lo) / 2;
                launch_sum_i32_i64(counts_.as<int32_t>() + lo, hi - lo, reinterpret_cast<int64_t*>(ctl_ + 4), s);
                PA_HIP(hipMemcpyAsync(h_ctl_ + 4, ctl_ + 4, 8, hipMemcpyDeviceToHost, s));
                PA_HIP(hipStreamSynchronize(s));
                memcpy(&sum, h_ctl_ + 4, 8);
            }
            if (sum > 0) break;
            // an empty leading range: move on
            lo = hi;
            hi = n;
            sum = remaining_;
        }
        const int32_t rows = hi - lo;
        const int32_t total = (int32_t)sum;
        last_matches_ = total;
        int32_t* probe_idx = static_cast<int32_t*>(probe_idx_.ensure((size_t)total * 4));
        int32_t* build_pos = static_cast<int32_t*>(build_pos_.ensure((size_t)total * 4));
        uint8_t* visited = track_visited_ && !filter_ ? ls_->visited_positions(s) : nullptr;
        if (tiles_valid_ && lo == 0 && hi == n) {
            // every probe row emits at most one pair and the whole page comes out at once: the exclusive scan runs over the totals of tiles
            // of 1024 probe rows (a thousandth of the rows) and the pairs are placed tile by tile, ranked inside the workgroup (2^23-row
            // pages: 73 us of scan passes over the rows' counts before; 2^26 random probe keys 38.5 -> 48.7 G rows/s)
            const int64_t tiles = join_probe_tiles(n);
            int32_t* tile_offsets = tile_totals_.as<int32_t>();
            launch_exclusive_scan_i32(tile_offsets, tile_offsets, tiles, nullptr, scan_temp_.ensure(scan_temp_bytes(tiles)), s);
            launch_join_probe_emit_tiles(head_.as<int32_t>(), tile_offsets, n, probe_flags_, probe_idx, build_pos, visited, s);
            tiles_valid_ = false;
        }
        else {
            tiles_valid_ = false;   // (a range of the page: the row counts are the ones to scan)
            int32_t* offsets = counts_.as<int32_t>() + lo;
            launch_exclusive_scan_i32(offsets, offsets, rows, nullptr, scan_temp_.ensure(scan_temp_bytes(rows)), s);
            launch_join_probe_emit(head_.as<int32_t>() + lo, offsets, rows, total, ls_->links.as<int32_t>(), probe_idx, build_pos, probe_flags_, visited, s);
        }
        int32_t total_out = total;
        if (filter_) {
            // the pairs so far are the CANDIDATES: every position of every probe row's chain; the filter decides which are joined
            total_out = apply_filter(lo, rows, total, probe_idx, build_pos, s);
            probe_idx = filtered_probe_.as<int32_t>();
            build_pos = filtered_build_.as<int32_t>();
            last_matches_ = total_out;
            if (total_out == 0) {  // nothing of this range survives: on to the next range (or done)
                range_lo_ = hi;
                remaining_ -= sum;
                pending_ = hi < n && remaining_ > 0;
                return pending_ ? get_output(out) : false;
            }
        }
        // LookupJoinPageBuilder.build: probe output channels by probe index (relative to the range) ++ build output channels
        // by build position
        // (the flat channels of both sides by ONE launch -- a page's gathers are small and their launches dominate --, strings one by one)
        size_t oc = 0;
        GatherMultiArgs gm;
        memset(&gm, 0, sizeof gm);
        gm.positions[0] = probe_idx;
        gm.positions[1] = build_pos;
        gm.count = total_out;
        auto flat = [&](int32_t type, const void* values, const uint8_t* nulls, int which, bool null_rows, OutColumn& out) {
            const int w = type_width(type);
            if ((w != 1 && w != 4 && w != 8) || gm.ncols >= GATHER_MULTI_MAX_COLS) return false;
            out.type = type;
            out.varwidth = false;
            out.is_view = false;
            out.host_ready = false;
            out.has_nulls = nulls != nullptr || null_rows;
            GatherMultiCol& g = gm.col[gm.ncols++];
            g.src = values;
            g.src_nulls = nulls;
            g.dst = out.values.ensure((size_t)std::max(total_out, 1) * w);
            g.dst_nulls = out.has_nulls ? static_cast<uint8_t*>(out.nulls.ensure((size_t)std::max(total_out, 1))) : nullptr;
            g.width = w;
            g.which = which;
            return true;
        };
        for (int c : output_channels_) {
            const DevColumn& src = in_.cols[c];
            const void* values = src.varwidth ? src.values : static_cast<const char*>(src.values) + (size_t)lo * type_width(src.type);
            OutColumn& out_col = out_cols_[oc++];
            if (!src.varwidth && flat(src.type, values, src.nulls ? src.nulls + lo : nullptr, 0, false, out_col)) continue;
            gather_column(src.type, src.varwidth, values, src.offsets ? src.offsets + lo : nullptr, src.nulls ? src.nulls + lo : nullptr, probe_idx, total_out,
                          out_col, s);
        }
        for (int c : ls_->output_channels) {
            const BuildColumn& src = ls_->cols[c];
            OutColumn& out_col = out_cols_[oc++];
            if (!src.varwidth && flat(src.type, src.values.ptr(), src.has_nulls ? src.nulls.as<uint8_t>() : nullptr, 1, probe_outer_, out_col)) continue;
            gather_column(src.type, src.varwidth, src.values.ptr(), src.offsets.as<int32_t>(), src.has_nulls ? src.nulls.as<uint8_t>() : nullptr,
                          build_pos, total_out, out_col, s, probe_outer_);
        }
        launch_gather_multi(gm, s);
        publish_output(out_cols_, total_out, output_mem_, s, out, out_storage_);
        range_lo_ = hi;
        remaining_ -= sum;
        pending_ = hi < n && remaining_ > 0;
        return true;
    }

    // A lookup source without duplicate keys joins every probe row to at most one build row: a page of n probe rows emits at most n
    // rows, so the pairs and the output columns can be sized for n and enqueued right behind the counting pass -- the match total is
    // read by the kernels on the device (tile offsets' grand total) and by the host only once, when get_output hands the page over.
    // One round trip per probe page instead of two (BenchmarkHashBuildAndJoinOperators' 1.4 M-row pages: kernels 50 us, round trips 25 us each).
    bool eager_possible(int32_t n) const
    {
        if (filter_ || track_visited_ || (int64_t)n > max_output_rows()) return false;
        for (int c : output_channels_) {
            const int w = type_width(in_.cols[c].type);
            if (in_.cols[c].varwidth || (w != 1 && w != 4 && w != 8)) return false;
        }
        for (int c : ls_->output_channels) {
            const int w = type_width(ls_->cols[c].type);
            if (ls_->cols[c].varwidth || (w != 1 && w != 4 && w != 8)) return false;
        }
        return output_channels_.size() + ls_->output_channels.size() <= (size_t)GATHER_MULTI_MAX_COLS;
    }
    void emit_eagerly(int32_t n, hipStream_t s)
    {
        const int64_t tiles = join_probe_tiles(n);
        int32_t* tile_offsets = tile_totals_.as<int32_t>();
        int32_t* total_dev = static_cast<int32_t*>(eager_total_.ensure(16));
        launch_exclusive_scan_i32(tile_offsets, tile_offsets, tiles, total_dev, scan_temp_.ensure(scan_temp_bytes(tiles)), s);
        int32_t* probe_idx = static_cast<int32_t*>(probe_idx_.ensure((size_t)n * 4));
        int32_t* build_pos = static_cast<int32_t*>(build_pos_.ensure((size_t)n * 4));
        launch_join_probe_emit_tiles(head_.as<int32_t>(), tile_offsets, n, probe_flags_, probe_idx, build_pos, nullptr, s);
        tiles_valid_ = false;
        GatherMultiArgs gm;
        memset(&gm, 0, sizeof gm);
        gm.positions[0] = probe_idx;
        gm.positions[1] = build_pos;
        gm.count = n;
        gm.count_dev = total_dev;
        size_t oc = 0;
        auto flat = [&](int32_t type, const void* values, const uint8_t* nulls, int which, bool null_rows, OutColumn& out) {
            const int w = type_width(type);
            out.type = type;
            out.varwidth = false;
            out.is_view = false;
            out.host_ready = false;
            out.has_nulls = nulls != nullptr || null_rows;
            GatherMultiCol& g = gm.col[gm.ncols++];
            g.src = values;
            g.src_nulls = nulls;
            g.dst = out.values.ensure((size_t)std::max(n, 1) * w);
            g.dst_nulls = out.has_nulls ? static_cast<uint8_t*>(out.nulls.ensure((size_t)std::max(n, 1))) : nullptr;
            g.width = w;
            g.which = which;
        };
        for (int c : output_channels_) flat(in_.cols[c].type, in_.cols[c].values, in_.cols[c].nulls, 0, false, out_cols_[oc++]);
        for (int c : ls_->output_channels) {
            const BuildColumn& src = ls_->cols[c];
            flat(src.type, src.values.ptr(), src.has_nulls ? src.nulls.as<uint8_t>() : nullptr, 1, probe_outer_, out_cols_[oc++]);
        }
        launch_gather_multi(gm, s);
        eager_ = true;
    }

    // private copies of the probe page's channels this operator reads (in_ is repointed to them)
    void keep_input(hipStream_t s)
    {
        const int64_t n = in_.n;
        kept_.resize(in_.cols.size() * 3);
        for (size_t c = 0; c < in_.cols.size(); c++) {
            if (!needed_[c]) continue;
            DevColumn& col = in_.cols[c];
            if (col.values == nullptr) continue;
            if (col.varwidth) {
                int32_t ends[2] = {0, 0};
                PA_HIP(hipMemcpyAsync(&ends[0], col.offsets, 4, hipMemcpyDeviceToHost, s));
                PA_HIP(hipMemcpyAsync(&ends[1], col.offsets + n, 4, hipMemcpyDeviceToHost, s));
                PA_HIP(hipStreamSynchronize(s));
                const size_t bytes = (size_t)(ends[1] - ends[0]);
                char* v = static_cast<char*>(kept_[3 * c].ensure(bytes ? bytes : 1));
                if (bytes) PA_HIP(hipMemcpyAsync(v, static_cast<const char*>(col.values) + ends[0], bytes, hipMemcpyDeviceToDevice, s));
                int32_t* o = static_cast<int32_t*>(kept_[3 * c + 1].ensure((size_t)(n + 1) * 4));
                PA_HIP(hipMemcpyAsync(o, col.offsets, (size_t)(n + 1) * 4, hipMemcpyDeviceToDevice, s));
                col.values = v - ends[0];  // (the offsets stay absolute)
                col.offsets = o;
            }
            else {
                const size_t bytes = (size_t)n * type_width(col.type);
                void* v = kept_[3 * c].ensure(bytes ? bytes : 1);
                if (bytes) PA_HIP(hipMemcpyAsync(v, col.values, bytes, hipMemcpyDeviceToDevice, s));
                col.values = v;
            }
            if (col.nulls) {
                uint8_t* nl = static_cast<uint8_t*>(kept_[3 * c + 2].ensure((size_t)n));
                PA_HIP(hipMemcpyAsync(nl, col.nulls, (size_t)n, hipMemcpyDeviceToDevice, s));
                col.nulls = nl;
            }
        }
        PA_HIP(hipStreamSynchronize(s));
        in_volatile_ = false;
    }

    void finish() override { finishing_ = true; }
    bool is_finished() override { return finishing_ && !pending_; }
    int64_t memory_bytes() override { return (int64_t)(stager_.bytes() + head_.capacity() + counts_.capacity() + probe_idx_.capacity() + build_pos_.capacity()); }

    void last_pairs(const int32_t** probe_idx, const int32_t** build_pos, int32_t* count) const
    {
        *probe_idx = filter_ ? filtered_probe_.as<int32_t>() : probe_idx_.as<int32_t>();
        *build_pos = filter_ ? filtered_build_.as<int32_t>() : build_pos_.as<int32_t>();
        *count = last_matches_;
    }

    // ---- JoinFilterFunction ----------------------------------------------------------------------------------------------
    // The filter is an ordinary RowExpression over a (build row, probe row) pair (JoinFilterFunctionCompiler.java numbers the build
    // page's channels first, then the probe page's).  It runs as a FilterAndProject operator of this library over the PAIR PAGE of
    // a probe page's candidates -- the channels the filter reads, gathered by build position / probe position, plus the
    // candidates' own indices, which is all it projects: what comes out are the indices of the eligible candidates, in order.
    void setup_filter(const pa_expr& f, bool single_match)
    {
        OwnedExpr e = OwnedExpr::copy(f);
        PA_REQUIRE(e.root_type() == PA_BOOLEAN, PA_ERR_INVALID_ARGUMENT, "join filter must be BOOLEAN");
        const int nb = (int)ls_->cols.size();
        std::set<int32_t> used;
        e.collect_channels(&used);
        for (int32_t c : used) PA_REQUIRE(c >= 0 && c < nb + n_probe_channels_, PA_ERR_INVALID_ARGUMENT, "join filter references a channel outside the build and probe pages");
        filter_types_.clear();
        for (int c = 0; c < nb; c++) filter_types_.push_back(ls_->cols[c].type);
        for (int c = 0; c < n_probe_channels_; c++) filter_types_.push_back(probe_types_[c]);
        filter_types_.push_back(PA_INTEGER);  // the candidate's index
        filter_used_.assign(filter_types_.size(), false);
        for (int32_t c : used) {
            filter_used_[c] = true;
            if (c >= nb) needed_[c - nb] = true;  // probe channels the filter reads are staged like the output channels
        }
        pa_expr_node idx{};
        idx.kind = PA_EXPR_INPUT_REF;
        idx.type = PA_INTEGER;
        idx.channel = nb + n_probe_channels_;
        pa_expr projection{};
        projection.node_count = 1;
        projection.root = 0;
        projection.nodes = &idx;
        pa_filter_project_desc fp{};
        fp.input_channel_count = (int32_t)filter_types_.size();
        fp.input_types = filter_types_.data();
        fp.filter = &f;
        fp.projection_count = 1;
        fp.projections = &projection;
        fp.output_mem = PA_MEM_DEVICE;
        fp.stream = stream_.get();
        filter_.reset(make_filter_project(&fp));
        filter_single_match_ = single_match;
        // candidates are emitted unfiltered: every position of every chain, no NULL-extended rows
        probe_flags_ = 0;
        pair_cols_.resize(filter_types_.size());
    }

    // candidates (probe_idx, build_pos)[0 .. total) of the probe rows [lo, lo + rows) -> filtered_probe_ / filtered_build_; returns their count
    int32_t apply_filter(int32_t lo, int32_t rows, int32_t total, const int32_t* cand_probe, const int32_t* cand_build, hipStream_t s)
    {
        const int nb = (int)ls_->cols.size();
        std::vector<pa_column> cols(filter_types_.size());
        for (size_t c = 0; c < cols.size(); c++) {
            memset(&cols[c], 0, sizeof(pa_column));
            cols[c].type = filter_types_[c];
            cols[c].encoding = filter_types_[c] == PA_VARCHAR ? PA_VARWIDTH : PA_FLAT;
            if (!filter_used_[c] && (int)c != nb + n_probe_channels_) continue;
            OutColumn& oc = pair_cols_[c];
            if ((int)c < nb) {
                const BuildColumn& src = ls_->cols[c];
                gather_column(src.type, src.varwidth, src.values.ptr(), src.offsets.as<int32_t>(), src.has_nulls ? src.nulls.as<uint8_t>() : nullptr, cand_build, total, oc, s);
            }
            else if ((int)c < nb + n_probe_channels_) {
                const DevColumn& src = in_.cols[c - nb];
                const void* values = src.varwidth ? src.values : static_cast<const char*>(src.values) + (size_t)lo * type_width(src.type);
                gather_column(src.type, src.varwidth, values, src.offsets ? src.offsets + lo : nullptr, src.nulls ? src.nulls + lo : nullptr, cand_probe, total, oc, s);
            }
            else {
                oc.type = PA_INTEGER;
                oc.varwidth = false;
                oc.has_nulls = false;
                launch_iota_i32(static_cast<int32_t*>(oc.values.ensure((size_t)total * 4)), total, s);
            }
            cols[c].values = oc.values.ptr();
            cols[c].offsets = oc.varwidth ? oc.offsets.as<int32_t>() : nullptr;
            cols[c].nulls = oc.has_nulls ? oc.nulls.as<uint8_t>() : nullptr;
        }
        pa_page pairs{};
        pairs.position_count = total;
        pairs.channel_count = (int32_t)cols.size();
        pairs.columns = cols.data();
        pairs.mem = PA_MEM_DEVICE;
        filter_->add_input(&pairs);
        pa_page kept{};
        int32_t ne = 0;
        const int32_t* eligible = nullptr;
        if (filter_->get_output(&kept)) {
            ne = kept.position_count;
            eligible = static_cast<const int32_t*>(kept.columns[0].values);
        }
        // outputSingleMatch: the first eligible position of a row only (DefaultPageJoiner.java:276-278)
        if (filter_single_match_ && ne > 0) {
            int32_t* keep = static_cast<int32_t*>(jf_keep_.ensure((size_t)(ne + 1) * 4));
            launch_jf_first_of_row(eligible, ne, cand_probe, keep, s);
            launch_exclusive_scan_i32(keep, keep, ne, ctl_ + 8, scan_temp_.ensure(scan_temp_bytes(ne)), s);
            PA_HIP(hipMemcpyAsync(h_ctl_ + 8, ctl_ + 8, 4, hipMemcpyDeviceToHost, s));
            PA_HIP(hipStreamSynchronize(s));
            const int32_t kept_n = h_ctl_[8];
            int32_t* firsts = static_cast<int32_t*>(jf_first_.ensure((size_t)std::max(kept_n, 1) * 4));
            launch_jf_compact(eligible, ne, keep, kept_n, firsts, s);
            eligible = firsts;
            ne = kept_n;
        }
        int32_t out_n = ne;
        if (!probe_outer_) {
            int32_t* op = static_cast<int32_t*>(filtered_probe_.ensure((size_t)std::max(ne, 1) * 4));
            int32_t* ob = static_cast<int32_t*>(filtered_build_.ensure((size_t)std::max(ne, 1) * 4));
            if (ne > 0) {
                launch_gather_flat(cand_probe, 4, eligible, ne, op, s);
                launch_gather_flat(cand_build, 4, eligible, ne, ob, s);
            }
        }
        else {
            // a row without an eligible position comes out once, NULL-extended, in its place (DefaultPageJoiner.java:296-303)
            int32_t* per_row = static_cast<int32_t*>(jf_rows_.ensure((size_t)(rows + 1) * 4 * 3));
            int32_t* first = per_row + (rows + 1);
            int32_t* at = first + (rows + 1);
            PA_HIP(hipMemsetAsync(per_row, 0, (size_t)(rows + 1) * 4, s));
            launch_jf_count_rows(eligible, ne, cand_probe, per_row, s);
            launch_exclusive_scan_i32(per_row, first, rows, nullptr, scan_temp_.ensure(scan_temp_bytes(rows)), s);
            launch_jf_max1(per_row, rows, at, s);
            launch_exclusive_scan_i32(at, at, rows, ctl_ + 8, scan_temp_.ensure(scan_temp_bytes(rows)), s);
            PA_HIP(hipMemcpyAsync(h_ctl_ + 8, ctl_ + 8, 4, hipMemcpyDeviceToHost, s));
            PA_HIP(hipStreamSynchronize(s));
            out_n = h_ctl_[8];
            int32_t* op = static_cast<int32_t*>(filtered_probe_.ensure((size_t)std::max(out_n, 1) * 4));
            int32_t* ob = static_cast<int32_t*>(filtered_build_.ensure((size_t)std::max(out_n, 1) * 4));
            launch_jf_outer(eligible, ne, cand_probe, cand_build, per_row, first, at, rows, op, ob, s);
        }
        // OuterPositionTracker: only joined build rows count as visited
        if (track_visited_ && out_n > 0) launch_jf_mark_visited(filtered_build_.as<int32_t>(), out_n, ls_->visited_positions(s), s);
        return out_n;
    }

private:
    // null_rows: positions may hold -1 = a NULL row (build side of a probe-outer join)
    void gather_column(int32_t type, bool varwidth, const void* values, const int32_t* offsets, const uint8_t* nulls, const int32_t* positions,
                       int32_t count, OutColumn& oc, hipStream_t s, bool null_rows = false)
    {
        oc.type = type;
        oc.varwidth = varwidth;
        oc.is_view = false;
        oc.host_ready = false;
        oc.has_nulls = nulls != nullptr || null_rows;
        if (varwidth) {
            int32_t* lens = static_cast<int32_t*>(oc.offsets.ensure((size_t)(count + 1) * 4));
            launch_varwidth_lengths(positions, count, offsets, nulls, lens, s);
            launch_sum_i32_i64(lens, count, reinterpret_cast<int64_t*>(ctl_ + 6), s);
            launch_exclusive_scan_i32(lens, lens, count, ctl_ + 2, scan_temp_.ensure(scan_temp_bytes(count)), s);
            PA_HIP(hipMemcpyAsync(h_ctl_ + 2, ctl_ + 2, 4, hipMemcpyDeviceToHost, s));
            PA_HIP(hipMemcpyAsync(h_ctl_ + 6, ctl_ + 6, 8, hipMemcpyDeviceToHost, s));
            PA_HIP(hipStreamSynchronize(s));
            int64_t bytes64;
            memcpy(&bytes64, h_ctl_ + 6, 8);
            PA_REQUIRE(bytes64 < ((int64_t)1 << 31), PA_ERR_INSUFFICIENT_RESOURCES, "a join output page holds more than 2 GiB of VARCHAR bytes");
            int32_t bytes = h_ctl_[2];
            uint8_t* dst = static_cast<uint8_t*>(oc.values.ensure((size_t)(bytes > 0 ? bytes : 1)));
            launch_varwidth_copy(positions, count, offsets, static_cast<const uint8_t*>(values), nulls, lens, dst, ctl_ + 2, s);
        }
        else if (null_rows) {
            int w = type_width(type);
            launch_gather_or_null(values, w, nulls, positions, count, oc.values.ensure((size_t)count * w), static_cast<uint8_t*>(oc.nulls.ensure((size_t)count)), s);
            return;
        }
        else {
            int w = type_width(type);
            launch_gather_flat(values, w, positions, count, oc.values.ensure((size_t)count * w), s);
        }
        if (null_rows) launch_gather_or_null(nullptr, 0, nulls, positions, count, nullptr, static_cast<uint8_t*>(oc.nulls.ensure((size_t)count)), s);
        else if (nulls) launch_gather_nulls(nulls, positions, count, static_cast<uint8_t*>(oc.nulls.ensure((size_t)count)), s);
    }

    Stream stream_;
    PageStager stager_;
    std::shared_ptr<LookupSourceImpl> ls_;
    int n_probe_channels_ = 0;
    std::vector<int32_t> probe_types_;
    std::vector<int> join_channels_, output_channels_;
    std::vector<bool> needed_;
    int hash_channel_ = -1, output_mem_ = PA_MEM_HOST;
    DevPage in_;
    bool in_volatile_ = false;
    std::vector<DevBuf> kept_;  // keep_input: per channel values, offsets, NULL flags
    DevBuf ctl_buf_, hash_, head_, counts_, probe_idx_, build_pos_, scan_temp_, totals_;
    PinnedBuf h_totals_;
    bool totals_pending_ = false;
    bool eager_ = false;          // the page's pairs and output columns are already in the stream (emit_eagerly)
    DevBuf eager_total_;
    // JoinFilterFunction
    std::unique_ptr<pa_operator> filter_;
    bool filter_single_match_ = false;
    std::vector<int32_t> filter_types_;
    std::vector<bool> filter_used_;
    std::vector<OutColumn> pair_cols_;
    DevBuf filtered_probe_, filtered_build_, jf_keep_, jf_first_, jf_rows_;
    PinnedBuf h_ctl_buf_;
    int32_t* ctl_ = nullptr;
    int32_t* h_ctl_ = nullptr;
    int32_t last_matches_ = 0;
    int32_t range_lo_ = 0;     // first probe row not joined yet (a probe page may come out as several pages)
    int64_t remaining_ = -1;   // matches of the rows from range_lo_ on (-1: not read back yet)
    bool finishing_ = false, pending_ = false, probe_outer_ = false, track_visited_ = false;
    DevBuf tile_totals_;         // output rows per tile of the probe page (launch_join_probe_count_keyed)
    bool tiles_valid_ = false;   // ... of the page being joined, not yet turned into offsets
    int probe_flags_ = 0;
    std::vector<OutColumn> out_cols_;
    std::vector<pa_column> out_storage_;
};

// LookupOuterOperator (…/operator/join/LookupOuterOperator.java:40-215): after the probe operators of the bridge are
// finished, the build rows whose position was never appended to a join output, in ascending position
// (OuterLookupSource.SharedLookupOuterPositionIterator, OuterLookupSource.java:120-160); probe channels NULL.
class LookupOuterOperator : public pa_operator {
public:
    LookupOuterOperator(const pa_lookup_join_desc* d, pa_lookup_source* bridge) : stream_(d->stream)
    {
        PA_REQUIRE(d != nullptr, PA_ERR_INVALID_ARGUMENT, "descriptor is null");
        require_device();
        PA_REQUIRE(bridge->impl != nullptr, PA_ERR_ILLEGAL_STATE, "lookup source has no build operator yet");
        PA_REQUIRE(d->join_type == PA_JOIN_LOOKUP_OUTER || d->join_type == PA_JOIN_FULL_OUTER, PA_ERR_INVALID_ARGUMENT,
                   "LookupOuterOperator belongs to a lookup-outer or full-outer join");
        ls_ = bridge->impl;
        for (int i = 0; i < d->probe_output_channel_count; i++) {
            int c = d->probe_output_channels[i];
            PA_REQUIRE(c >= 0 && c < d->probe_channel_count, PA_ERR_INVALID_ARGUMENT, "probe output channel out of range");
            probe_output_types_.push_back(d->probe_types[c]);
        }
        output_mem_ = d->output_mem;
        ctl_ = static_cast<int32_t*>(ctl_buf_.ensure(64));
        h_ctl_ = static_cast<int32_t*>(h_ctl_buf_.ensure(64));
    }
    ~LookupOuterOperator() override { (void)hipStreamSynchronize(stream_.get()); }
    hipStream_t private_stream() override { return stream_.owned() ? stream_.get() : nullptr; }
    hipStream_t main_stream() override { return stream_.get(); }

    bool needs_input() override { return false; }  // LookupOuterOperator.java:135-138
    void add_input(const pa_page*) override { throw Error(PA_ERR_ILLEGAL_STATE, "LookupOuterOperator does not take input"); }
    bool is_blocked() override { return !ls_->built.load(); }
    void finish() override {}
    bool is_finished() override { return done_; }

    bool get_output(pa_page* out) override
    {
        if (done_ || !ls_->built.load()) return false;
        done_ = true;
        hipStream_t s = stream_.get();
        const int64_t n = ls_->n;
        if (n == 0) return false;
        int32_t* part = static_cast<int32_t*>(part_.ensure((size_t)n * 4));
        int32_t* pos = static_cast<int32_t*>(pos_.ensure((size_t)n * 4));
        int64_t* counts = static_cast<int64_t*>(counts_.ensure(64));
        launch_join_unvisited_flag(ls_->visited_positions(s), n, part, s);
        launch_partition_positions(part, n, 2, pos, counts, part_temp_.ensure(partition_temp_bytes(n, 2)), s);
        int64_t h_counts[2];
        PA_HIP(hipMemcpyAsync(h_counts, counts, 16, hipMemcpyDeviceToHost, s));
        PA_HIP(hipStreamSynchronize(s));
        const int32_t count = (int32_t)h_counts[0];
        if (count == 0) return false;
        out_cols_.clear();
        out_cols_.resize(probe_output_types_.size() + ls_->output_channels.size());
        size_t oc = 0;
        for (int32_t t : probe_output_types_) {  // appendNullForProbe: LookupOuterOperator.java:178-186
            OutColumn& o = out_cols_[oc++];
            o.type = t;
            o.varwidth = t == PA_VARCHAR;
            o.has_nulls = true;
            PA_HIP(hipMemsetAsync(o.nulls.ensure((size_t)count), 1, (size_t)count, s));
            if (o.varwidth) {
                PA_HIP(hipMemsetAsync(o.offsets.ensure((size_t)(count + 1) * 4), 0, (size_t)(count + 1) * 4, s));
                o.values.ensure(1);
            }
            else {
                PA_HIP(hipMemsetAsync(o.values.ensure((size_t)count * type_width(t)), 0, (size_t)count * type_width(t), s));
            }
        }
        for (int c : ls_->output_channels) {
            const BuildColumn& src = ls_->cols[c];
            OutColumn& o = out_cols_[oc++];
            o.type = src.type;
            o.varwidth = src.varwidth;
            const uint8_t* nulls = src.has_nulls ? src.nulls.as<uint8_t>() : nullptr;
            o.has_nulls = nulls != nullptr;
            if (src.varwidth) {
                int32_t* lens = static_cast<int32_t*>(o.offsets.ensure((size_t)(count + 1) * 4));
                launch_varwidth_lengths(pos, count, src.offsets.as<int32_t>(), nulls, lens, s);
                launch_exclusive_scan_i32(lens, lens, count, ctl_ + 2, scan_temp_.ensure(scan_temp_bytes(count)), s);
                PA_HIP(hipMemcpyAsync(h_ctl_ + 2, ctl_ + 2, 4, hipMemcpyDeviceToHost, s));
                PA_HIP(hipStreamSynchronize(s));
                const int32_t bytes = h_ctl_[2];
                launch_varwidth_copy(pos, count, src.offsets.as<int32_t>(), src.values.as<uint8_t>(), nulls, lens,
                                     static_cast<uint8_t*>(o.values.ensure((size_t)(bytes > 0 ? bytes : 1))), ctl_ + 2, s);
            }
            else {
                const int w = type_width(src.type);
                launch_gather_flat(src.values.ptr(), w, pos, count, o.values.ensure((size_t)count * w), s);
            }
            if (nulls) launch_gather_nulls(nulls, pos, count, static_cast<uint8_t*>(o.nulls.ensure((size_t)count)), s);
        }
        publish_output(out_cols_, count, output_mem_, s, out, out_storage_);
        return true;
    }

private:
    Stream stream_;
    std::shared_ptr<LookupSourceImpl> ls_;
    std::vector<int32_t> probe_output_types_;
    int output_mem_ = PA_MEM_HOST;
    bool done_ = false;
    DevBuf ctl_buf_, part_, pos_, counts_, part_temp_, scan_temp_;
    PinnedBuf h_ctl_buf_;
    int32_t* ctl_ = nullptr;
    int32_t* h_ctl_ = nullptr;
    std::vector<OutColumn> out_cols_;
    std::vector<pa_column> out_storage_;
};

}  // namespace

bool lookup_source_key_bitmap(pa_lookup_source* ls, const uint64_t** bits, int64_t* min_key, uint64_t* range, std::shared_ptr<void>* keep)
{
    PA_REQUIRE(ls != nullptr && ls->impl, PA_ERR_INVALID_ARGUMENT, "lookup source is null");
    PA_REQUIRE(ls->impl->built.load(), PA_ERR_ILLEGAL_STATE, "the lookup source is not built yet");
    if (!ls->impl->bitmap.bits) return false;
    *bits = ls->impl->bitmap.bits;
    *min_key = ls->impl->bitmap.min_key;
    *range = ls->impl->bitmap.range;
    *keep = ls->impl;
    return true;
}

// For a dynamic filter that spans the ranks of a partitioned join (every rank holds the build keys of its partition): the
// local key range, and the local keys as bits of a caller-provided bitmap over a common range.
bool lookup_source_key_range(pa_lookup_source* ls, int64_t* min_key, int64_t* max_key)
{
    PA_REQUIRE(ls != nullptr && ls->impl, PA_ERR_INVALID_ARGUMENT, "lookup source is null");
    PA_REQUIRE(ls->impl->built.load(), PA_ERR_ILLEGAL_STATE, "the lookup source is not built yet");
    if (!ls->impl->keyed || !ls->impl->key_range_valid) return false;
    *min_key = ls->impl->key_min;
    *max_key = ls->impl->key_max;
    return true;
}
void lookup_source_fill_bitmap(pa_lookup_source* ls, int64_t min_key, uint64_t range, uint64_t* bits, hipStream_t s)
{
    PA_REQUIRE(ls != nullptr && ls->impl && bits != nullptr, PA_ERR_INVALID_ARGUMENT, "null argument");
    PA_REQUIRE(ls->impl->built.load() && ls->impl->keyed, PA_ERR_ILLEGAL_STATE, "needs a built lookup source with one integer join key");
    const JoinKeys bk = ls->impl->build_keys();
    launch_join_key_bitmap(bk.col[0], ls->impl->n, min_key, range, bits, s);
}

// The dynamic filter of a partitioned join: every rank holds the build keys of its own partition, and the probe side is
// filtered BEFORE it is exchanged, so the filter must know every rank's keys.  Collective over the ranks of `comm`.
bool lookup_source_shared_bitmap(pa_lookup_source* ls, pa_comm* comm, bool partitioned_by_key, hipStream_t s, const uint64_t** bits, int64_t* min_key,
                                 uint64_t* range)
{
    PA_REQUIRE(ls != nullptr && ls->impl, PA_ERR_INVALID_ARGUMENT, "lookup source is null");
    LookupSourceImpl& impl = *ls->impl;
    PA_REQUIRE(impl.built.load(), PA_ERR_ILLEGAL_STATE, "the lookup source is not built yet");
    const int64_t big = (int64_t)1 << 62;
    const bool has = impl.keyed && impl.key_range_valid;
    // [min key, -max key, "this rank cannot take part"] -> MIN / MIN / MAX; rows -> SUM
    int64_t ends[2] = {has ? impl.key_min : big, has ? -impl.key_max : big};
    int64_t flags[1] = {impl.keyed ? 0 : 1};
    int64_t rows[1] = {impl.n};
    comm_all_reduce_i64(comm, ends, 2, COMM_MIN, s);
    comm_all_reduce_i64(comm, flags, 1, COMM_MAX, s);
    comm_all_reduce_i64(comm, rows, 1, COMM_SUM, s);
    if (flags[0] != 0 || ends[0] == big) return false;  // not one integer key everywhere, or no key anywhere
    const int64_t lo = ends[0], hi = -ends[1];
    const uint64_t r = (uint64_t)(hi - lo);
    // a bitmap pays while it is no larger than the keys themselves (the rule of the single-rank filter) and stays below 8 GiB
    if (hi < lo || r >= 64ULL * (uint64_t)std::max<int64_t>(rows[0], 1) || r >= (1ULL << 36)) return false;
    const int64_t words = (int64_t)(r >> 6) + 1;
    uint64_t* b = static_cast<uint64_t*>(impl.shared_bits.ensure((size_t)words * 8));
    if (has) {
        const JoinKeys bk = impl.build_keys();
        launch_join_key_bitmap(bk.col[0], impl.n, lo, r, b, s);
    }
    else PA_HIP(hipMemsetAsync(b, 0, (size_t)words * 8, s));
    if (comm->world > 1) {
        if (partitioned_by_key) {
            // every key lives on exactly one rank: the ranks' set bits are disjoint, SUM == OR (RCCL has no bitwise reduction)
            comm_all_reduce_sum_u64(comm, b, words, s);
        }
        else {
            DevBuf all;
            const int W = comm->world;
            uint64_t* g = static_cast<uint64_t*>(all.ensure((size_t)words * 8 * W));
            std::vector<int64_t> soff((size_t)W, 0), sb((size_t)W, words * 8), roff((size_t)W), rb((size_t)W, words * 8);
            for (int p = 0; p < W; p++) roff[p] = (int64_t)p * words * 8;
            comm_all_to_all_v(comm, b, soff.data(), sb.data(), g, roff.data(), rb.data(), s);
            launch_or_words(b, g, words, W, s);
            PA_HIP(hipStreamSynchronize(s));  // `all` goes back to the pool
        }
    }
    PA_HIP(hipStreamSynchronize(s));
    *bits = b;
    *min_key = lo;
    *range = r;
    return true;
}

pa_operator* make_hash_builder(const pa_hash_builder_desc* desc, pa_lookup_source* bridge)
{
    return new HashBuilderOperator(desc, bridge);
}
pa_operator* make_lookup_join(const pa_lookup_join_desc* desc, pa_lookup_source* bridge)
{
    return new LookupJoinOperator(desc, bridge);
}
pa_operator* make_lookup_outer(const pa_lookup_join_desc* desc, pa_lookup_source* bridge)
{
    return new LookupOuterOperator(desc, bridge);
}
pa_lookup_source* lookup_source_new() { return new pa_lookup_source(); }
void lookup_source_delete(pa_lookup_source* ls) { delete ls; }

int32_t lookup_join_last_pairs(pa_operator* op, const int32_t** probe_idx, const int32_t** build_pos, int32_t* count)
{
    auto* j = dynamic_cast<LookupJoinOperator*>(op);
    PA_REQUIRE(j != nullptr, PA_ERR_INVALID_ARGUMENT, "not a LookupJoin operator");
    j->last_pairs(probe_idx, build_pos, count);
    return PA_OK;
}

int32_t lookup_source_position_count(pa_lookup_source* ls)
{
    PA_REQUIRE(ls != nullptr && ls->impl != nullptr && ls->impl->built.load(), PA_ERR_ILLEGAL_STATE, "lookup source is not built");
    return ls->impl->n;
}

// key[] / positionLinks[] of a built lookup source (tests: chain order parity with the reference)
int32_t lookup_source_tables(pa_lookup_source* ls, const int32_t** key, int32_t* hash_size, const int32_t** links, int32_t* positions)
{
    PA_REQUIRE(ls != nullptr && ls->impl != nullptr && ls->impl->built.load(), PA_ERR_ILLEGAL_STATE, "lookup source is not built");
    LookupSourceImpl& impl = *ls->impl;
    if (!impl.reference_built) {
        // keyed joins probe their own table; PagesHash.key[] is made here, for whoever wants to compare it with the reference.
        // The chains are the ones in use: both constructions leave a key's highest position as head and link downwards.
        const int32_t n = impl.n;
        const JoinKeys bk = impl.build_keys();
        DevBuf slot_scratch, link_scratch, err;
        fill_raw_hash(impl, bk, n, nullptr);
        impl.key.ensure((size_t)(impl.mask + 1) * 4);
        int32_t* e = static_cast<int32_t*>(err.ensure(64));
        PA_HIP(hipMemsetAsync(e, 0, 64, nullptr));
        launch_join_build(bk, impl.raw_hash.as<int64_t>(), n, impl.key.as<int32_t>(), impl.mask,
                          static_cast<int32_t*>(slot_scratch.ensure((size_t)std::max(n, 1) * 4)),
                          static_cast<int32_t*>(link_scratch.ensure((size_t)std::max(n, 1) * 4)), e, nullptr);
        PA_HIP(hipStreamSynchronize(nullptr));
        impl.reference_built = true;
    }
    *key = ls->impl->key.as<int32_t>();
    *hash_size = (int32_t)(ls->impl->mask + 1);
    *links = ls->impl->links.as<int32_t>();
    *positions = ls->impl->n;
    return PA_OK;
}

}  // namespace pa
