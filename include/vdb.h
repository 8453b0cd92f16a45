/*
 * vdb.h -- C ABI of the MI355X brute-force similarity engine (libvdb.so).
 *
 * The library stands where the reference reaches its vector engine: the seven
 * methods /root/reference/app.py calls on a pymilvus `Milvus` object
 * (connect, has_table, create_table, create_index, delete_table, insert,
 * search_vectors).  Each entry point below names the reference call it replaces.
 *
 * Conventions
 *   - every function returns VDB_OK (0) on success, matching the reference's
 *     `status.code == 0` test (app.py:243), or a negative VDB_E* code;
 *     vdb_last_error() gives a thread-local message for the last failure
 *     (the reference prints `status.message`, app.py:245);
 *   - plain pointers and sizes only; the caller allocates every output;
 *   - host entry points take host pointers; `_dev` entry points take device
 *     pointers valid on the library's device and a HIP stream handle
 *     (a `hipStream_t` passed as `void*`, NULL = the default stream); the search
 *     and merge `_dev` calls enqueue their work on that stream and return
 *     without synchronising, vdb_insert_dev waits for its stream before it
 *     returns (the new rows are then visible to searches on any stream);
 *   - thread safety: any number of concurrent searches and inserts.  Inserts on a
 *     table are serialised among themselves and do not stall searches: new rows are
 *     written behind the published row count and become visible with one atomic
 *     store once the whole call has landed, so a search sees all rows of an insert
 *     or none; tables grow in place (reserved virtual ranges), nothing a search
 *     reads ever moves.  Only vdb_drop_table waits for the work in flight
 *     (the reference serves each HTTP request on its own thread against one
 *     process-global engine handle with no locking: app.py:11-12,186-190);
 *   - there is no CPU fallback: without a usable GPU vdb_init fails.
 */
#ifndef VDB_H
#define VDB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* return codes */
#define VDB_OK            0
#define VDB_EINVAL      (-1)   /* bad argument */
#define VDB_ENOTFOUND   (-2)   /* no such table */
#define VDB_EEXISTS     (-3)   /* table already exists */
#define VDB_ENOGPU      (-4)   /* no usable HIP device / runtime failure at init */
#define VDB_EHIP        (-5)   /* a HIP call failed */
#define VDB_ENOMEM      (-6)   /* HBM budget (gpu_resource_config.cache_capacity) or allocation exceeded */
#define VDB_ESTATE      (-7)   /* library not initialised / already shut down */

/* metric: the reference fixes MetricType.IP per table (app.py:22,141); L2 and COSINE are additions */
#define VDB_METRIC_L2     1    /* squared Euclidean distance, smaller is better */
#define VDB_METRIC_IP     2    /* inner product, larger is better */
#define VDB_METRIC_COSINE 3    /* rows L2-normalised on insert, query on search, then IP */

/* storage dtype of a table's rows in HBM */
#define VDB_F32   0
#define VDB_BF16  1
#define VDB_F16   2

/* index types accepted (and ignored: every search is an exhaustive scan) by vdb_create_index;
 * the reference asks for IndexType.IVFLAT, nlist 2048 (app.py:145-149) */
#define VDB_INDEX_FLAT    1
#define VDB_INDEX_IVFLAT  2

#define VDB_MAX_TOPK   1024   /* largest k accepted by the search entry points */

/* ---- lifetime: replaces Milvus() + connect(host, port) (app.py:171-182) ---------------- */
/* Binds the calling process to HIP device `device` (ordinal in the process's visible set).
 * Calling it again with the same device is a no-op. */
int vdb_init(int device);
int vdb_shutdown(void);
/* Knobs of /root/reference/server_config.yaml, forwarded by the host shim after it has read
 * that file: "use_blas_threshold" (:36, batch size from which the MFMA GEMM path is considered; default 2),
 * "cache_capacity_gb" (:42, HBM budget for all tables, 0 = unlimited),
 * "insert_buffer_size_gb" (:19, size of the device staging buffer host inserts are chunked through).
 * Launch tunables of the scan kernel (defaults are the measured best): "scan_nt",
 * "scan_waves", "scan_blocks_per_cu", "scan_contiguous"; "scan_serialize" (1 = scans issued on
 * different streams run one after the other, chained by events, while the short kernels around
 * them overlap; 0 = scans of different streams may overlap); "scan_sample" (1 = for long candidate
 * lists over short rows, a scan of 1/256 of the rows first supplies a threshold floor to the full scan;
 * rows of up to 4096 bytes, or of up to the value given when it is larger than 1; 0 = never);
 * "scan_dynamic" (1 = the waves of a scan workgroup draw their steps from a counter in LDS (default);
 * 0 = every wave scans a fixed share); "scan_short_gb" (tables below this many GB are scanned by one resident
 * round of workgroups whose waves advance side by side; default 28); "fin_threads" (threads of the finalize
 * workgroup behind a scan: 512 (default), 256 or 1024; raised by itself when a thread would own more than eight lists).
 * Batched path (all its passes run on the 16-bit matrix unit; the finalize step re-scores every
 * candidate within twice the error bound of the k-th best fast key, so a coarser pass costs
 * re-scoring, never exactness): "gemm_path" (0 sends every batch through the scan path),
 * "gemm32_split" (fp32 tables: 2 = rows and queries rounded to bf16, one product per term (default);
 * 1 = bf16 hi + lo parts, three products per term; 0 = fp32 MFMA),
 * "gemm_split_bf16" / "gemm_split_f16" (1 = the fp32 query goes in as hi + lo parts of the table's
 * type, two products per term; default 0: rounded once),
 * "gemm32s_threads" (workgroup shape of the 256-query tile: 512 = 8 waves over 256 rows (default),
 * 256 = 4 waves over 128 rows, two workgroups per CU).
 * "gemm_pp" (1 = batches of more than 128 queries run on the ping-pong kernel that loads its
 * operands straight into LDS (default); 0 = the converting kernel for every batch size),
 * "gemm_pp_tile" (rows per tile of that kernel: 0 = 384 x 256 tiles for launches of up to 256 queries and
 * 256 x 256 tiles beyond (default); 256 or 384 = that shape for every launch),
 * "gemm_pp_parts" (1 = the last pass of that kernel tightens its threshold after an eighth of the
 * rows (default)).
 * Store: "bf16_mirror" (1 = fp32 tables keep a bf16 copy of their rows, half of the table again in
 * HBM, which the batched passes multiply from (default); 0 = no copy, rows are converted on the fly),
 * "virtual_ranges" (1 = table arrays are reserved virtual ranges backed chunk by chunk as the table
 * grows, so growth never copies (default); 0 = plain allocations, copied on growth),
 * "scan_mirror" (1 = single queries over fp32 tables scan the bf16 mirror: half the bytes, the same
 * exact answers through a wider certificate; a query it cannot certify is re-scanned over the fp32
 * rows before anything goes to the exact pass ("scan_mirror_tier2", default 1, switches that off);
 * default 0).  The three apply to tables created, and searches started, after the call.
 * Exactness layer: "l2_row_bounds" (1 = L2 keys carry the error term of their own row, so rows of very
 * large norm do not send other queries to the exact pass (default); 0 = table-wide norm bound),
 * "ip_outliers" (1 = IP tables keep rows whose norm is 8 to 11 times the typical one, up to 1024 of them, on
 * a side list: the batched path takes its error bound over the other rows and re-scores the listed rows
 * canonically for every query (default); 0 = table-wide bound; applies to rows inserted after the call),
 * "force_exact" (1 = every query is answered by the exact pass; for measuring it).
 * Host calls: "coalesce_host" (1 = blocking vdb_search calls that arrive from other threads while one
 * is running are merged into one batched search, see vdb_search, and vdb_insert calls of up to 256 rows
 * into one append; default 1), "host_direct" (1 = blocking vdb_search calls of up to four queries leave out both
 * copies: the kernels read the queries from the library's pinned host block and write the answers into it, and over
 * tables under 256 MB the caller watches one word per query there instead of waiting for the stream, for at most
 * 200 us; default 1; 0 = copy in, copy out, wait for the stream). */
int vdb_set_option(const char* key, double value);
const char* vdb_last_error(void);

/* ---- table lifecycle ---------------------------------------------------------------------- */
/* replaces create_table({'table_name','dimension','index_file_size','metric_type'}) (app.py:137-143) */
int vdb_create_table(const char* name, int64_t dim, int metric, int dtype);
/* replaces has_table(name) -> (status, bool) (app.py:120) */
int vdb_has_table(const char* name, int* out_exists);
/* replaces delete_table(name) (app.py:242) */
int vdb_drop_table(const char* name);
/* replaces create_index(name, {'index_type','nlist'}) (app.py:145-149): validates and returns */
int vdb_create_index(const char* name, int index_type, int nlist);
/* row count of a table (pymilvus get_table_row_count; not called by app.py, used by tests/bench) */
int vdb_count(const char* name, int64_t* out_rows);
/* grow the table's capacity to at least `rows` without changing its contents */
int vdb_reserve(const char* name, int64_t rows);
/* describe a table: any out pointer may be NULL */
int vdb_table_info(const char* name, int64_t* dim, int* metric, int* dtype,
                   int64_t* rows, int64_t* capacity, int64_t* row_stride_bytes);

/* HBM bytes the bf16 mirror of an fp32 table occupies at the moment (see option "bf16_mirror"); 0 for
 * 16-bit tables, for tables created without one, and for tables whose mirror was dropped: when rows
 * no longer fit beside the mirror (HBM exhausted, or gpu_resource_config.cache_capacity,
 * /root/reference/server_config.yaml:42), an insert, reserve or load drops the mirror of that table
 * (a notice goes to stderr) and goes on, instead of failing with VDB_ENOMEM; batched searches over the
 * table then convert its fp32 rows on the fly, with the same exact answers. */
int vdb_table_mirror_bytes(const char* name, int64_t* out_bytes);

/* Number of tables created in this process so far (vdb_create_table and vdb_load_table).  It only grows.
 * A binding that keeps the dimension of a table beside its name (to check the shape of a query block
 * without a vdb_table_info call per search: describe_table, app.py:203-205, is what the reference asks
 * before every lookup) forgets what it kept when this number has moved: only a creation can make a
 * name stand for a table of another shape.  No counterpart in the reference's client. */
int64_t vdb_tables_created(void);

/* ---- append: replaces insert(table, records=[vec], ids=[id]) (app.py:315-325) ------------- */
/* rows: [n, dim] row-major fp32 on the host; ids: [n] caller-supplied signed 64-bit ids.
 * COSINE tables normalise each row on the device while appending. */
int vdb_insert(const char* name, int64_t n, const float* rows, const int64_t* ids);
/* same with rows/ids already in device memory */
int vdb_insert_dev(const char* name, int64_t n, const float* rows_dev, const int64_t* ids_dev,
                   void* stream);
/* append n synthetic rows generated on the device: element (row, col) of the raw fp32 row is a
 * pure function of (seed, first_row + i, col); ids are first_row + i.  The ingestion-side
 * counterpart of /root/reference/createVector.py:6-8 for sizes that cannot pass through HTTP. */
int vdb_insert_synthetic(const char* name, int64_t n, uint64_t seed, int64_t first_row);
/* fill a caller-owned device buffer [n, dim] fp32 with the same generator (queries, tests) */
int vdb_synth_fill_dev(float* out_dev, int64_t n, int64_t dim, uint64_t seed, int64_t first_row,
                       void* stream);
/* copy stored rows [first, first+n) back to the host as fp32 (tests and snapshots);
 * out_rows [n, dim], out_ids [n]; either may be NULL */
int vdb_fetch_rows(const char* name, int64_t first, int64_t n, float* out_rows, int64_t* out_ids);

/* ---- snapshots: the engine-side part of db_config.primary_path / preload_table
 * (/root/reference/server_config.yaml:12, :22), which the Milvus server the reference talks to
 * uses to keep tables across restarts and to bring them into memory at start-up.
 * vdb_save_table writes the stored (quantised, for COSINE normalised) rows, ids and row norms of
 * one table to `path`.  The first save of a table to a path writes the whole table under a temporary
 * name and renames it.  A later save of the same table (same identity: a table that was created, or
 * loaded from that file, and has only grown since) to the same path is incremental: it appends one
 * segment with the rows that arrived since the last save and rewrites the header last, so the file
 * reads as the older snapshot until the new rows are complete; the cost is that of the new rows.
 * The reference's engine persists every insert under primary_path, and its /insert relies on that
 * (a hash already in the relational table is never sent to the engine again, app.py:293-295);
 * incremental saves are what makes frequent snapshots affordable here.
 * vdb_save_table_rows is the same call and also reports how many rows it wrote.
 * vdb_load_table creates table `name` from such a file: VDB_EEXISTS if the name is taken,
 * VDB_ENOTFOUND if the file cannot be opened, VDB_EINVAL if it is not a complete snapshot.  A loaded
 * table answers every search with the same ids and scores as the table that was saved. */
int vdb_save_table(const char* name, const char* path);
int vdb_save_table_rows(const char* name, const char* path, int64_t* out_rows_written);
int vdb_load_table(const char* name, const char* path);

/* ---- search: replaces search_vectors(table_name, query_records, top_k, nprobe) (app.py:384-391)
 * queries: [nq, dim] fp32.  Outputs, per query q: out_counts[q] = min(k, rows) hits, best first,
 * ties ordered by id ascending; out_ids[q*k + i], out_scores[q*k + i]; unused slots hold id -1.
 * Scores: IP / COSINE similarity (larger first) or squared L2 distance (smaller first).
 * nprobe of the reference call has no counterpart: the scan is exhaustive.
 * vdb_search blocks.  The reference serves every request on its own thread against one engine
 * handle (app.py:11-12, 186-190), so calls arrive side by side: a call that arrives while another
 * one is running waits for it, and the next call to run takes along every waiting call on the same
 * table with the same k (up to 1024 queries) as one batched search.  Results are the same as for
 * separate calls; a lone caller is never delayed. */
int vdb_search(const char* name, int64_t nq, const float* queries, int k,
               int64_t* out_ids, float* out_scores, int32_t* out_counts);
int vdb_search_dev(const char* name, int64_t nq, const float* queries_dev, int k,
                   int64_t* out_ids_dev, float* out_scores_dev, int32_t* out_counts_dev,
                   void* stream);

/* No counterpart in the reference.  The same search as vdb_search_dev for callers that cannot
 * allocate device memory (tight serving loops, harnesses that forbid it): the call allocates,
 * creates and frees nothing (no device memory, stream or event).  Its intermediate buffers live in a 24 MiB
 * array inside the library's code object (each GPU the library runs on holds its copy until the
 * process ends); a search that needs more (many queries per call, a large k: about 3.1 MiB per query
 * at k = 10 on a large table, 12.6 MiB at k = 100) is refused with VDB_EINVAL, as are more than 256
 * queries per call.  Every query runs as a scan (no batched path); results are those of
 * vdb_search_dev bit for bit.  All calls on one GPU share the array: a call made while another
 * thread is inside the entry point is refused with VDB_ESTATE; calls on one stream are kept apart by
 * stream order; calls on different streams must be ordered by the caller. */
int vdb_search_dev_noalloc(const char* name, int64_t nq, const float* queries_dev, int k,
                           int64_t* out_ids_dev, float* out_scores_dev, int32_t* out_counts_dev,
                           void* stream);

/* ---- row-sharded tables: merge of per-shard results (no counterpart in the reference, which
 * talks to one engine: server_config.yaml:8 deploy_mode single) -------------------------------
 * in_*: [nparts, nq, k] as produced by vdb_search_dev on each shard and gathered by the host
 * side (RCCL all-gather); writes the global top-k in the same format. */
int vdb_merge_topk_dev(int nparts, int64_t nq, int k, int metric,
                       const int64_t* in_ids_dev, const float* in_scores_dev,
                       const int32_t* in_counts_dev,
                       int64_t* out_ids_dev, float* out_scores_dev, int32_t* out_counts_dev,
                       void* stream);

/* Per-rank result record for one collective: ids [nq*k] int64 | scores [nq*k] fp32 | counts [nq]
 * int32, padded to a multiple of 16 bytes.  A rank points vdb_search_dev's three outputs into its
 * record, the records are all-gathered back to back, and vdb_merge_records_dev merges them. */
int64_t vdb_record_bytes(int64_t nq, int k);
int vdb_merge_records_dev(int nparts, int64_t nq, int k, int metric, const void* records_dev,
                          int64_t* out_ids_dev, float* out_scores_dev, int32_t* out_counts_dev,
                          void* stream);

/* ---- measurement hooks (bench.py) ------------------------------------------------------------ */
/* on = 1: every launch of the dominant kernel (table scan, full batched pass) is bracketed by HIP
 * events on its stream; on = n > 1: every n-th launch (a pair of timing events costs about 10 us
 * between two kernels, which is several per cent of a scan over a short table); 0: off. */
int vdb_prof_enable(int on);
/* Synchronises the recorded events; returns summed scan-kernel milliseconds and launch count
 * since the last call, and resets both. */
int vdb_prof_read(double* scan_ms, int64_t* scan_launches);
/* Reads `bytes` of a device buffer with the scan's access shape and no scoring: the read-bandwidth
 * ceiling the scan's achieved GB/s is compared with (tools/tune_scan.py). */
int vdb_bench_read_dev(const void* buf_dev, int64_t bytes, int nt, int blocks, int threads, void* stream);
/* counters since init: queries answered, queries that needed the exact second pass, queries
 * that went through the batched (MFMA GEMM) path */
int vdb_stats(int64_t* queries, int64_t* exact_fallbacks, int64_t* batched_queries);

#ifdef __cplusplus
}
#endif
#endif /* VDB_H */
