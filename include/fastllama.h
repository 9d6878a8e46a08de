/*
 * fastllama.h -- PRIMARY drop-in boundary of libfastllama_hip.so: the `llama_*` C-ABI that the reference's
 * Python class (interfaces/python/fastllama.py, ctypes) and C examples bind.
 *
 * Every declaration below is ABI-identical to /root/reference/interfaces/c/fastllama.h (cited per item): same
 * symbol names, same struct layouts (ctypes mirrors them field for field, fastllama.py:132-151), same ownership
 * and error conventions (bool results, -1 for perplexity, borrowed array views, callbacks get (ptr, len) that
 * are not NUL-terminated).  What differs is what sits behind them: Model::eval runs device-resident on an
 * MI355X (include/fastllama_hip.h, "the model"); tokenizer, sampler, batching and session logic stay on the
 * host in C++ as in the reference (lib/bridge.cpp).  `n_threads`, `use_mmap`, `use_mlock`, `load_parallel`,
 * `n_load_parallel_blocks` and `allocate_extra_mem` are accepted and ignored (they tune the CPU executor).
 *
 * To switch an application: point fastllama.Model(library_path=...) (fastllama.py:236) at this library.
 */
#ifndef FASTLLAMA_AMD_FASTLLAMA_H
#define FASTLLAMA_AMD_FASTLLAMA_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden: these declarations ARE its export list */

/* reference: `enum progress_type_tag : uint8_t`, fastllama.h:12-20 (passed as one byte) */
typedef uint8_t progress_type_tag;
enum {
    PROGRESS_TAG_UNKNOWN = 0,
    PROGRESS_TAG_INIT = 1,
    PROGRESS_TAG_LOAD = 2,
    PROGRESS_TAG_SAVE = 3,
    PROGRESS_TAG_INGEST = 4,
    PROGRESS_TAG_ATTACH_LORA_ADAPTER = 5,
    PROGRESS_TAG_DETACH_LORA_ADAPTER = 6
};

/* fastllama.h:22-25 */
typedef void (*LLAMA_LOGGER_FUNC)(char const *function_name, int function_name_size, char const *message, int message_size);
typedef void (*LLAMA_LOGGER_RESET_FUNC)(void);
typedef void (*LLAMA_LOGGER_PROGRESS_FUNC)(progress_type_tag, size_t done_size, size_t total_size);
typedef void (*LLAMA_STREAM_FUNC)(char const *token_stream, int token_stream_size);

struct llama_model_context;

/* fastllama.h:30-36 */
struct llama_logger {
    LLAMA_LOGGER_FUNC log;
    LLAMA_LOGGER_FUNC log_err;
    LLAMA_LOGGER_FUNC log_warn;
    LLAMA_LOGGER_RESET_FUNC reset;
    LLAMA_LOGGER_PROGRESS_FUNC progress;
};

/* fastllama.h:39-42 -- a BORROWED view, valid until the next eval / reset / free */
struct llama_array_view_f {
    float const *data;
    size_t size;
};

/* fastllama.h:46-61 */
struct llama_model_context_args {
    bool embedding_eval_enabled;
    bool should_get_all_logits;
    bool use_mmap;
    bool use_mlock;
    bool load_parallel;
    int seed;
    int n_keep;
    int n_ctx;
    int n_threads;
    int n_batch;
    uint32_t n_load_parallel_blocks;
    size_t last_n_tokens;
    size_t allocate_extra_mem;
    struct llama_logger logger;
};

struct llama_model_context_args llama_create_default_context_args(void);                 /* fastllama.h:64  */
struct llama_model_context *llama_create_context(struct llama_model_context_args args);  /* fastllama.h:72  */
bool llama_load_model(struct llama_model_context *ctx, char const *filepath);            /* fastllama.h:82  */
bool llama_set_stop_words(struct llama_model_context *ctx, char const **words, size_t len); /* :93 */
bool llama_ingest_system_prompt(struct llama_model_context *ctx, char const *prompt);    /* fastllama.h:104 */
bool llama_ingest(struct llama_model_context *ctx, char const *prompt);                  /* fastllama.h:115 */
bool llama_generate(struct llama_model_context *ctx, LLAMA_STREAM_FUNC stream_fn, size_t number_of_tokens, float top_k,
                    float top_p, float temp, float repeat_penalty);                      /* fastllama.h:131-139 */
float llama_perplexity(struct llama_model_context *ctx, char const *prompt);             /* fastllama.h:148, -1 on failure */
struct llama_array_view_f llama_get_embeddings(struct llama_model_context const *ctx);   /* fastllama.h:157 */
struct llama_array_view_f llama_get_logits(struct llama_model_context const *ctx);       /* fastllama.h:166 */
bool llama_save_state(struct llama_model_context *ctx, char const *filepath);            /* fastllama.h:175 */
bool llama_load_state(struct llama_model_context *ctx, char const *filepath);            /* fastllama.h:184 */
bool llama_attach_lora(struct llama_model_context *ctx, char const *filepath);           /* fastllama.h:194 */
bool llama_detach_lora(struct llama_model_context *ctx);                                 /* fastllama.h:203 */
bool llama_reset_model(struct llama_model_context *ctx);                                 /* fastllama.h:212 */
void llama_free_context(struct llama_model_context *ctx);                                /* fastllama.h:218 */
void llama_handle_signal(int);                        /* exported but undeclared in the reference, c/main.cpp:229 */

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif /* FASTLLAMA_AMD_FASTLLAMA_H */
