// llama_api.cpp -- the PRIMARY drop-in boundary: the 17 `llama_*` symbols of the reference's C-ABI
// (/root/reference/interfaces/c/fastllama.h:64-218 + llama_handle_signal, interfaces/c/main.cpp:229),
// re-implemented on top of the device-resident model (fl_model, model.cpp).
//
// Host-side pieces written here (C++, as in the reference), each mirroring the behaviour of:
//   model file reader (GGML / GGMF v1 / GGJT v1)   include/file_loader.hpp:94-250      -> load_model_file
//   vocabulary + SentencePiece-style tokenizer     include/tokenizer.hpp:71-176        -> tokenize
//   sampler (repeat penalty, top-k, top-p, temp)   lib/bridge.cpp:13-108               -> sample_top_p_top_k
//   streaming token buffer with stop words         include/token_buffer.hpp            -> TokenBuffer
//   session: ingest / generate / perplexity / state lib/bridge.cpp:110-559             -> Session
// Model::eval itself (lib/llama.cpp:272-499) is fl_model_eval on the GPU.  There is no CPU evaluation path:
// if no gfx950 device is present llama_load_model fails and says so.
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <queue>
#include <random>
#include <sstream>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/fastllama.h"
#include "../../include/fastllama_hip.h"

namespace {

using token_t = int32_t;
constexpr token_t TOKEN_BOS = 1, TOKEN_EOS = 2;   // include/bridge.hpp:18-19

// ------------------------------------------------------------------------------------------------ logger
struct Log {
    llama_logger cb{};
    void emit(LLAMA_LOGGER_FUNC f, const char *fn, const std::string &msg) const {
        if (f) f(fn, (int)strlen(fn), msg.data(), (int)msg.size());
    }
    void info(const char *fn, const std::string &m) const { emit(cb.log, fn, m); }
    void err(const char *fn, const std::string &m) const { emit(cb.log_err, fn, m); }
    void warn(const char *fn, const std::string &m) const { emit(cb.log_warn, fn, m); }
    void reset() const { if (cb.reset) cb.reset(); }
    void progress(progress_type_tag t, size_t done, size_t total) const { if (cb.progress) cb.progress(t, done, total); }
};

void def_log_info(const char *fn, int fl, const char *m, int ml) { printf("\x1b[32;1m[Info]:\x1b[0m Func('%.*s') %.*s", fl, fn, ml, m); fflush(stdout); }
void def_log_err(const char *fn, int fl, const char *m, int ml) { fprintf(stderr, "\x1b[31;1m[Error]:\x1b[0m Func('%.*s') %.*s", fl, fn, ml, m); fflush(stderr); }
void def_log_warn(const char *fn, int fl, const char *m, int ml) { printf("\x1b[93;1m[Warn]:\x1b[0m Func('%.*s') %.*s", fl, fn, ml, m); fflush(stdout); }
void def_log_reset() { printf("\x1b[0m"); fflush(stdout); }
void def_log_progress(progress_type_tag, size_t, size_t) {}
// the operator library's warnings (a derived operand copy that did not fit: fl_set_warn_handler) go to the logger of the most recently
// loaded session, as the reference's own warnings do
Log g_warn_log;
void forward_warning(const char *line) { g_warn_log.warn("fastllama_hip", std::string(line) + "\n"); }

// ------------------------------------------------------------------------------------------------ vocabulary
struct Vocab {
    std::vector<std::string> tok;
    std::vector<float> score;
    std::unordered_map<std::string, token_t> to_id;
    std::string_view token(token_t id) const {
        return (size_t)id < tok.size() ? std::string_view(tok[(size_t)id]) : std::string_view{};
    }
};

// SentencePiece-style greedy bigram merging by vocabulary score (include/tokenizer.hpp:71-176): split into UTF-8
// characters, repeatedly merge the adjacent pair whose concatenation is the highest-scoring vocabulary entry
// (ties: leftmost), emit ids, bytes of unknown pieces fall back to id = byte + 3.
size_t utf8_len(char c) {
    static const size_t lut[16] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 2, 2, 3, 4};
    return lut[(uint8_t)c >> 4];
}

std::vector<token_t> tokenize(const Vocab &v, std::string_view text, bool bos) {
    std::vector<token_t> out;
    if (text.empty()) return out;
    if (bos) out.push_back(TOKEN_BOS);
    struct Sym { size_t pos, len; int prev, next; };
    struct Bigram { int left, right; float score; size_t size; };
    struct Cmp { bool operator()(const Bigram &l, const Bigram &r) const { return l.score < r.score || (l.score == r.score && l.left > r.left); } };
    std::vector<Sym> syms;
    for (size_t off = 0; off < text.size();) {
        const size_t n = std::min(text.size() - off, utf8_len(text[off]));
        Sym s{off, n, (int)syms.size() - 1, -1};
        off += n;
        s.next = off == text.size() ? -1 : (int)syms.size() + 1;
        syms.push_back(s);
    }
    std::priority_queue<Bigram, std::vector<Bigram>, Cmp> q;
    auto try_add = [&](int l, int r) {
        if (l == -1 || r == -1) return;
        const std::string piece(text.substr(syms[l].pos, syms[l].len + syms[r].len));
        auto it = v.to_id.find(piece);
        if (it == v.to_id.end() || (size_t)it->second >= v.tok.size()) return;
        q.push(Bigram{l, r, v.score[(size_t)it->second], piece.size()});
    };
    for (size_t i = 1; i < syms.size(); ++i) try_add((int)i - 1, (int)i);
    while (!q.empty()) {
        const Bigram b = q.top();
        q.pop();
        Sym &L = syms[b.left], &R = syms[b.right];
        if (L.len == 0 || R.len == 0 || L.len + R.len != b.size) continue;
        L.len += R.len;
        L.next = R.next;
        if (R.next >= 0) syms[R.next].prev = b.left;
        R.len = 0;
        try_add(L.prev, b.left);
        try_add(b.left, L.next);
    }
    for (int i = 0; i != -1; i = syms[i].next) {
        const std::string piece(text.substr(syms[i].pos, syms[i].len));
        auto it = v.to_id.find(piece);
        if (it != v.to_id.end()) out.push_back(it->second);
        else for (char c : piece) out.push_back((token_t)((uint8_t)c) + 3);
    }
    return out;
}

// ------------------------------------------------------------------------------------------------ sampler
// lib/bridge.cpp:24-108.  temp <= 0: argmax.  Otherwise logits/temp with the CTRL repetition penalty on the last-n
// tokens, top-k (partial sort), softmax in double over exp(float), top-p cut, std::discrete_distribution on mt19937.
token_t sample_top_p_top_k(const float *logits, int n_vocab, const std::deque<token_t> &last_n, double repeat_penalty,
                           int top_k, double top_p, double temp, std::mt19937 &rng) {
    if (temp <= 0.0) return (token_t)(std::max_element(logits, logits + n_vocab) - logits);
    std::vector<std::pair<double, token_t>> lid((size_t)n_vocab);
    const std::unordered_set<token_t> recent(last_n.begin(), last_n.end());
    const double scale = 1.0 / temp, inv_pen = 1.0 / repeat_penalty;
    for (int i = 0; i < n_vocab; ++i) {
        const double sl = (double)logits[i] * scale;
        lid[(size_t)i] = {recent.count(i) ? sl * (logits[i] < 0.0f ? repeat_penalty : inv_pen) : sl, i};
    }
    const int k = top_k > 0 ? std::min(top_k, n_vocab) : n_vocab;
    std::partial_sort(lid.begin(), lid.begin() + k, lid.end(), [](const auto &a, const auto &b) { return a.first > b.first; });
    lid.resize((size_t)k);
    const double maxl = lid[0].first;
    std::vector<double> probs(lid.size());
    double sum = 0.0;
    for (size_t i = 0; i < lid.size(); ++i) {
        probs[i] = (double)std::exp((float)(lid[i].first - maxl));
        sum += probs[i];
    }
    for (double &p : probs) p /= sum;
    if (top_p < 1.0) {
        double cum = 0.0;
        for (size_t i = 0; i < probs.size(); ++i) {
            cum += probs[i];
            if (cum >= top_p) {
                probs.resize(i + 1);
                lid.resize(i + 1);
                break;
            }
        }
    }
    std::discrete_distribution<> dist(probs.begin(), probs.end());
    return lid[(size_t)dist(rng)].second;
}

// ------------------------------------------------------------------------------------------------ token buffer
// include/token_buffer.hpp: holds back as many tokens as the longest stop word has, so a stop word can be cut out
// before it is streamed; incomplete UTF-8 tails wait for their continuation bytes.
struct TokenBufferState {
    std::string left_out;
    std::string unicode_backlog;
};

struct TokenBuffer {
    const Vocab &vocab;
    size_t max_size;
    std::function<void(const std::string &)> fn;
    std::deque<token_t> buf;
    std::string backlog;

    void fix_utf8(std::string &s) {
        if (s.empty()) return;
        if (!backlog.empty()) { s = backlog + s; backlog.clear(); }
        size_t last = 0, ulen = 0;
        for (size_t i = 0; i < s.size();) { ulen = utf8_len(s[i]); last = i; i += ulen; }
        if (last + ulen > s.size()) { backlog = s.substr(last); s.resize(last); }
    }
    void flush_one() {
        if (buf.empty()) return;
        std::string t(vocab.token(buf.front()));
        buf.pop_front();
        fix_utf8(t);
        if (!t.empty()) fn(t);
    }
    void add(token_t id) {
        if (max_size <= buf.size()) flush_one();
        buf.push_back(id);
    }
    // -> (found, text before the stop word, text after it)
    bool find_stop(const std::vector<std::string> &stops, std::string &before, std::string &after) const {
        if (stops.empty()) return false;
        std::string s = backlog;
        for (token_t id : buf) s += vocab.token(id);
        for (const auto &w : stops) {
            const size_t p = s.find(w);
            if (p != std::string::npos) { before = s.substr(0, p); after = s.substr(p + w.size()); return true; }
        }
        return false;
    }
    void restore(TokenBufferState &st) {
        if (!st.left_out.empty()) {
            fix_utf8(st.left_out);
            if (!st.left_out.empty()) fn(st.left_out);
            st.left_out.clear();
        }
        backlog = st.unicode_backlog;
    }
};

// ------------------------------------------------------------------------------------------------ model file
struct MappedFile {
    const uint8_t *p = nullptr;
    size_t n = 0;
    int fd = -1;
    bool open(const char *path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        n = (size_t)st.st_size;
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) return false;
        p = (const uint8_t *)m;
        return true;
    }
    ~MappedFile() {
        if (p) munmap((void *)p, n);
        if (fd >= 0) close(fd);
    }
};

struct Reader {
    const uint8_t *p;
    size_t n, off = 0;
    bool ok = true;
    template <class T> T get() {
        T v{};
        if (off + sizeof(T) > n) { ok = false; return v; }
        memcpy(&v, p + off, sizeof(T));
        off += sizeof(T);
        return v;
    }
    std::string str(size_t len) {
        if (off + len > n) { ok = false; return {}; }
        std::string s((const char *)p + off, len);
        off += len;
        return s;
    }
};

struct HParams { int32_t n_vocab = 0, n_embd = 0, n_mult = 0, n_head = 0, n_layer = 0, n_rot = 0, ftype = 0; };

// ------------------------------------------------------------------------------------------------ session
struct Session {
    llama_model_context_args args{};
    Log log;
    HParams hp;
    Vocab vocab;
    fl_model *model = nullptr;
    int max_batch = 0;
    int n_past = 0, seed = 0, keep = 0;
    size_t mem_per_token = 0;
    std::mt19937 rng;
    std::vector<token_t> embd;
    std::deque<token_t> last_n;      // RingBuffer<token_id_t>, include/ring_buffer.hpp:26-40
    size_t last_n_cap = 64;
    std::vector<float> logits, embeddings;
    int logits_on_device = 0;   // > 0: the host copy of that many logits rows is still owed (perplexity keeps them in HBM)
    std::vector<token_t> system_prompt;
    TokenBufferState tb_state;
    bool all_logits = false;

    ~Session() { if (model) fl_model_free(model); }

    void push_last(token_t t) {
        if (last_n.size() >= last_n_cap) last_n.pop_front();
        last_n.push_back(t);
    }

    // One model file: header, vocab and the tensor directory (file_loader.hpp:37-252).
    struct TensorEntry { std::string name; uint32_t type = 0, n_dims = 0, ne[2] = {1, 1}; size_t off = 0, bytes = 0; };
    struct ModelFile {
        MappedFile f;
        HParams hp;
        int version = 0;                                             // 0 GGML, 1 GGMF v1, 2 GGJT v1
        std::vector<TensorEntry> tensors;
    };

    bool parse_file(const std::string &path, ModelFile &mf, bool want_vocab) {
        if (!mf.f.open(path.c_str())) { log.err("Model::load", "unable to open '" + path + "'\n"); return false; }
        Reader r{mf.f.p, mf.f.n};
        const uint32_t magic = r.get<uint32_t>();
        if (magic == 0x67676d6c) mf.version = 0;
        else if (magic == 0x67676d66 || magic == 0x67676a74) {
            const uint32_t fv = r.get<uint32_t>();
            if (fv != 1) { log.err("read_magic_number", "unsupported file version\n"); return false; }
            mf.version = magic == 0x67676d66 ? 1 : 2;
        } else { log.err("read_magic_number", "invalid model file '" + path + "' (bad magic)\n"); return false; }
        HParams &h = mf.hp;
        h.n_vocab = r.get<int32_t>(); h.n_embd = r.get<int32_t>(); h.n_mult = r.get<int32_t>(); h.n_head = r.get<int32_t>();
        h.n_layer = r.get<int32_t>(); h.n_rot = r.get<int32_t>(); h.ftype = r.get<int32_t>();
        if (!r.ok || h.n_vocab <= 0 || h.n_embd <= 0 || h.n_head <= 0 || h.n_layer <= 0 || h.n_mult <= 0) {
            log.err("read_hyperparams", "failed to read hyper parameters\n");
            return false;
        }
        // sizes derived from these fields are allocated before any tensor is seen: refuse what no LLaMA file holds
        // (a vocabulary entry takes at least 4 bytes of the file)
        if (h.n_embd > (1 << 16) || h.n_head > h.n_embd || h.n_layer > 1024 || h.n_mult > (1 << 16) ||
            (size_t)h.n_vocab > (r.n - r.off) / 4) {
            log.err("read_hyperparams", "implausible hyper parameters (corrupt file?)\n");
            return false;
        }
        if (want_vocab) {
            vocab.tok.resize((size_t)h.n_vocab);
            vocab.score.assign((size_t)h.n_vocab, 0.f);
        }
        for (int i = 0; i < h.n_vocab; ++i) {
            const uint32_t len = r.get<uint32_t>();
            std::string w = r.str(len);
            const float sc = mf.version >= 1 ? r.get<float>() : 0.f;
            if (!r.ok) { log.err("read_vocab", "failed to read vocab\n"); return false; }
            if (want_vocab) {
                vocab.score[(size_t)i] = sc;
                vocab.to_id[w] = i;
                vocab.tok[(size_t)i] = std::move(w);
            }
        }
        while (r.ok && r.off < r.n) {
            TensorEntry t;
            t.n_dims = r.get<uint32_t>();
            const uint32_t name_len = r.get<uint32_t>();
            t.type = r.get<uint32_t>();
            if (!r.ok) break;
            if (t.n_dims < 1 || t.n_dims > 2) { log.err("read_tensor_metadata", "tensor has a bad number of dimensions\n"); return false; }
            for (uint32_t d = 0; d < t.n_dims; ++d) t.ne[d] = r.get<uint32_t>();
            t.name = r.str(name_len);
            if (!r.ok || t.ne[0] == 0 || t.ne[1] == 0) { log.err("read_tensor_metadata", "tensor '" + t.name + "' has an empty or truncated shape\n"); return false; }
            if (mf.version >= 2) r.off += (size_t)(-(int64_t)r.off & 31);
            const size_t nel = (size_t)t.ne[0] * t.ne[1];
            if (t.type == 0) t.bytes = nel * 4;
            else if (t.type == 1) t.bytes = nel * 2;
            else if (t.type == FL_TYPE_Q4_0) t.bytes = nel / 32 * 20;
            else if (t.type == FL_TYPE_Q4_1) t.bytes = nel / 32 * 24;
            else { log.err("read_tensor_metadata", "unrecognized tensor type\n"); return false; }
            if (!r.ok || r.off + t.bytes > r.n) { log.err("read_tensor_metadata", "truncated tensor '" + t.name + "'\n"); return false; }
            t.off = r.off;
            r.off += t.bytes;
            mf.tensors.push_back(std::move(t));
        }
        return true;
    }

    // Model::load (lib/llama.cpp:105-270) + ModelLoader (file_loader.hpp:377-644).  Multi-part checkpoints
    // (<path>, <path>.1, ...; count = n_embd / ne0 of tok_embeddings in the first part, :443-453) are merged the way the
    // loader's split table intends -- tok_embeddings / wo / w2 by columns (each row is the concatenation of the parts'
    // rows), every other matrix by rows, vectors not split (tensor/utils.hpp:93-112) -- and go to HBM as one tensor.
    bool load(const char *path) {
        std::vector<std::unique_ptr<ModelFile>> files;
        files.emplace_back(new ModelFile());
        if (!parse_file(path, *files[0], true)) return false;
        hp = files[0]->hp;
        if (hp.ftype != FL_TYPE_Q4_0 && hp.ftype != FL_TYPE_Q4_1) {
            log.err("Model::load", "this build evaluates Q4_0 / Q4_1 models on the GPU (file ftype " + std::to_string(hp.ftype) + ")\n");
            return false;
        }
        size_t n_files = 1;
        for (const TensorEntry &t : files[0]->tensors)
            if (t.name == "tok_embeddings.weight" && t.ne[0] > 0) n_files = (size_t)hp.n_embd / t.ne[0];
        if (n_files < 1) n_files = 1;
        for (size_t i = 1; i < n_files; ++i) {
            files.emplace_back(new ModelFile());
            const std::string pi = std::string(path) + "." + std::to_string(i);
            if (!parse_file(pi, *files[i], false)) return false;
            if (memcmp(&files[i]->hp, &hp, sizeof hp) != 0) {
                log.err("ModelLoader", "Hyper parameters mismatch between '" + pi + "' and '" + path + "'\n");
                return false;
            }
        }
        const int n_ff = ((2 * (4 * hp.n_embd) / 3 + hp.n_mult - 1) / hp.n_mult) * hp.n_mult;   // lib/llama.cpp:129
        max_batch = std::min(args.n_ctx, std::max(args.n_batch, args.n_keep + (int)args.last_n_tokens + args.n_batch));
        fl_model_params mp{};
        mp.n_vocab = hp.n_vocab; mp.n_embd = hp.n_embd; mp.n_head = hp.n_head; mp.n_layer = hp.n_layer; mp.n_ff = n_ff;
        mp.n_ctx = args.n_ctx; mp.qtype = hp.ftype; mp.max_batch = max_batch; mp.tp_rank = 0; mp.tp_size = 1;
        if (fl_init(0) != FL_OK) { log.err("Model::load", std::string(fl_last_error()) + "\n"); return false; }
        model = fl_model_create(&mp);
        if (!model) { log.err("Model::load", std::string(fl_last_error()) + "\n"); return false; }
        log.info("Model::load", "n_vocab=" + std::to_string(hp.n_vocab) + " n_embd=" + std::to_string(hp.n_embd) + " n_head=" +
                                    std::to_string(hp.n_head) + " n_layer=" + std::to_string(hp.n_layer) + " n_ff=" + std::to_string(n_ff) +
                                    (n_files > 1 ? " parts=" + std::to_string(n_files) : std::string()) + "\n");
        size_t done = 0;
        std::vector<uint8_t> merged;
        for (const TensorEntry &t0 : files[0]->tensors) {
            const uint8_t *data = files[0]->f.p + t0.off;
            uint32_t ne0 = t0.ne[0], ne1 = t0.ne[1];
            if (n_files > 1 && t0.n_dims == 2) {
                std::vector<const TensorEntry *> sh{&t0};
                for (size_t i = 1; i < n_files; ++i) {
                    const TensorEntry *found = nullptr;
                    for (const TensorEntry &t : files[i]->tensors) if (t.name == t0.name) found = &t;
                    if (!found || found->type != t0.type || found->ne[0] != t0.ne[0] || found->ne[1] != t0.ne[1] || found->bytes != t0.bytes) {
                        log.err("ModelLoader", "inconsistent tensor shards for '" + t0.name + "'\n");
                        return false;
                    }
                    sh.push_back(found);
                }
                const bool by_cols = t0.name.rfind("tok_embeddings.", 0) == 0 || t0.name.find(".attention.wo.weight") != std::string::npos ||
                                     t0.name.find(".feed_forward.w2.weight") != std::string::npos;
                merged.resize(t0.bytes * n_files);
                if (by_cols) {
                    const size_t row_bytes = t0.bytes / t0.ne[1];
                    for (uint32_t rrow = 0; rrow < t0.ne[1]; ++rrow)
                        for (size_t i = 0; i < n_files; ++i)
                            memcpy(merged.data() + ((size_t)rrow * n_files + i) * row_bytes, files[i]->f.p + sh[i]->off + (size_t)rrow * row_bytes, row_bytes);
                    ne0 = t0.ne[0] * (uint32_t)n_files;
                } else {
                    for (size_t i = 0; i < n_files; ++i) memcpy(merged.data() + i * t0.bytes, files[i]->f.p + sh[i]->off, t0.bytes);
                    ne1 = t0.ne[1] * (uint32_t)n_files;
                }
                data = merged.data();
            }
            if (fl_model_set_tensor(model, t0.name.c_str(), (int)t0.type, data, (int)ne0, (int)ne1) != FL_OK) {
                log.err("Model::load", std::string(fl_last_error()) + "\n");
                return false;
            }
            log.progress(PROGRESS_TAG_LOAD, ++done, (size_t)(3 + 9 * hp.n_layer));
        }
        if (fl_model_finalize(model) != FL_OK) { log.err("Model::load", std::string(fl_last_error()) + "\n"); return false; }
        log.info("Model::load", "model resident in HBM: " + std::to_string(fl_model_device_bytes(model) >> 20) + " MiB\n");
        return true;
    }

    // ---- LoRA (lib/llama.cpp:697-944): "ggla" v1 file = { u8 use_cache_matrix, u32 r, u32 alpha } + tensors named
    // <base>.loraA [r, ne0(base)] (pre-scaled by alpha/r, scripts/convert-lora-to-ggml.py:140-145) and <base>.loraB
    // [r, ne1(base)], or <base>.lora [ne0, ne1] (the cached product).  Every pair is merged into the resident weights by
    // fl_model_lora_apply (dequantize + add + re-quantize, the reference's add_q_f32).  use_mmap keeps the originals for
    // an exact detach (:714-726, :868-874); otherwise detach re-applies the file with the sign flipped (:921-938).
    std::string attached_lora;

    bool apply_lora_file(const std::string &path, float sign, bool keep_backup, const char *fn, progress_type_tag tag) {
        MappedFile f;
        if (!f.open(path.c_str())) { log.err("FileLoader", "Failed to open file: '" + path + "'\n"); return false; }
        Reader r{f.p, f.n};
        const uint32_t magic = r.get<uint32_t>(), fv = r.get<uint32_t>();
        if (!r.ok || magic != 0x67676c61 || fv != 1) { log.err("read_magic_number", "invalid model file " + path + " (bad magic)\n"); return false; }
        const bool use_cache = r.get<uint8_t>() != 0;
        const uint32_t rank = r.get<uint32_t>(), alpha = r.get<uint32_t>();
        if (!r.ok) { log.err(fn, "truncated lora adapter header\n"); return false; }
        if (sign > 0) {
            char buf[32];
            snprintf(buf, sizeof buf, "%.2f", rank ? (double)alpha / (double)rank : 0.0);
            log.info(fn, "lora_params: \n");
            log.info(fn, std::string("   Use cached matrix= ") + (use_cache ? "Yes" : "No") + "\n");
            log.info(fn, "   Alpha            = " + std::to_string(alpha) + "\n");
            log.info(fn, "   Rank             = " + std::to_string(rank) + "\n");
            log.info(fn, std::string("   Scale            = ") + buf + "\n");
        }
        struct Half { const float *p = nullptr; uint32_t ne0 = 0, ne1 = 0; };
        std::map<std::string, std::pair<Half, Half>> pending;           // base -> (A, B)
        bool warned = false;
        size_t done = 0;
        while (r.ok && r.off < r.n) {
            const uint32_t n_dims = r.get<uint32_t>(), name_len = r.get<uint32_t>(), type = r.get<uint32_t>();
            if (!r.ok) break;
            if (n_dims != 2) { log.err(fn, "lora adapter only supports matrices\n"); return false; }
            const uint32_t ne0 = r.get<uint32_t>(), ne1 = r.get<uint32_t>();
            const std::string name = r.str(name_len);
            r.off += (size_t)(-(int64_t)r.off & 31);
            const size_t nel = (size_t)ne0 * ne1, bytes = type == 0 ? nel * 4 : type == 1 ? nel * 2 : 0;
            if (!r.ok || bytes == 0 || r.off + bytes > r.n) { log.err(fn, "bad or truncated tensor '" + name + "' in lora adapter\n"); return false; }
            const float *data = reinterpret_cast<const float *>(r.p + r.off);
            r.off += bytes;
            const std::string suffix = ".lora";
            const size_t pos = name.rfind(suffix);
            const size_t want_pos = name.size() - suffix.size() - (use_cache ? 0 : 1);
            if (pos == std::string::npos || pos != want_pos) { log.err(fn, "'" + name + "' is not a lora tensor\n"); return false; }
            const std::string base = name.substr(0, pos);
            int b0 = 0, b1 = 0;
            if (fl_model_lora_shape(model, base.c_str(), &b0, &b1) != FL_OK) { log.err(fn, "unknown tensor '" + base + "' in lora adapter\n"); return false; }
            if (type != 0) { log.err(fn, "currently, we support fp32 lora tensors only.\n"); return false; }
            if (!warned) {
                log.warn(fn, "using a lora adapter with a quantized model may result in poor quality, use a f16 or f32 base model\n");
                warned = true;
            }
            int rc = FL_OK;
            bool merged = false;
            if (use_cache) {
                if ((int)ne0 != b0 || (int)ne1 != b1) {
                    log.err(fn, "incompatible tensor dimensions (" + std::to_string(b0) + " and " + std::to_string(ne1) + ") are you sure that this adapter is for this model?\n");
                    return false;
                }
                rc = fl_model_lora_apply(model, base.c_str(), data, nullptr, nullptr, 0, sign, keep_backup ? 1 : 0);
                merged = true;
            } else {
                auto &pr = pending[base];
                Half &h = name.back() == 'A' ? pr.first : pr.second;
                h.p = data; h.ne0 = ne0; h.ne1 = ne1;
                if (pr.first.p && pr.second.p) {
                    if ((int)pr.first.ne1 != b0 || (int)pr.second.ne1 != b1 || pr.first.ne0 != pr.second.ne0) {
                        log.err(fn, "incompatible tensor dimensions (" + std::to_string(b0) + " and " + std::to_string(pr.first.ne1) + ") are you sure that this adapter is for this model?\n");
                        return false;
                    }
                    rc = fl_model_lora_apply(model, base.c_str(), nullptr, pr.first.p, pr.second.p, (int)pr.first.ne0, sign, keep_backup ? 1 : 0);
                    pending.erase(base);
                    merged = true;
                }
            }
            if (merged && rc != FL_OK) { log.err(fn, std::string(fl_last_error()) + "\n"); return false; }
            done += nel;
            log.progress(tag, done, f.n / 4);
        }
        if (!r.ok) { log.err(fn, "failed to load all tensors\n"); return false; }
        return true;
    }

    bool attach_lora(const char *path) {
        if (!attached_lora.empty()) {
            log.err("attach_lora", "already attached LoRa model from '" + attached_lora + "'. Detach it first or reload the model.\n");
            return false;
        }
        log.info("attach_lora", std::string("attaching LoRa model from '") + path + "'. Please wait ...\n");
        if (!apply_lora_file(path, 1.0f, args.use_mmap, "attach_lora", PROGRESS_TAG_ATTACH_LORA_ADAPTER)) {
            if (args.use_mmap) (void)fl_model_lora_restore(model);     // leave the model as it was
            return false;
        }
        attached_lora = path;
        return true;
    }

    bool detach_lora() {
        if (attached_lora.empty()) { log.err("detach_lora", "no LoRa model attached.\n"); return false; }
        log.info("detach_lora", "detaching LoRa model from '" + attached_lora + "'. Please wait ...\n");
        bool ok;
        if (args.use_mmap) {
            ok = fl_model_lora_restore(model) == FL_OK;
            if (!ok) log.err("detach_lora", std::string(fl_last_error()) + "\n");
        } else {
            ok = apply_lora_file(attached_lora, -1.0f, false, "detach_lora", PROGRESS_TAG_DETACH_LORA_ADAPTER);
        }
        if (ok || args.use_mmap) attached_lora.clear();
        return ok;
    }

    // Model::eval(n_past, tokens, logits, ...) -- lib/llama.cpp:272; batches larger than max_batch are split
    // the logits of the last eval on the host (llama_get_logits, save_state): perplexity() leaves them in HBM
    void sync_logits() {
        if (logits_on_device <= 0) return;
        logits.resize((size_t)hp.n_vocab * (size_t)logits_on_device);
        if (fl_model_logits_read(model, 0, logits_on_device, logits.data()) != FL_OK) logits.clear();
        logits_on_device = 0;
    }

    bool eval(int past, const std::vector<token_t> &toks, bool keep_on_device = false) {
        const int V = hp.n_vocab, N = (int)toks.size();
        if (N == 0) return true;
        logits_on_device = 0;
        if (keep_on_device && all_logits && N <= max_batch) {      // one device eval, no host copy yet
            if (fl_model_eval(model, toks.data(), N, past, nullptr, 0, args.embedding_eval_enabled ? (embeddings.resize((size_t)hp.n_embd), embeddings.data()) : nullptr) != FL_OK) {
                log.err("Model::eval", std::string(fl_last_error()) + "\n");
                return false;
            }
            logits_on_device = N;
            if (mem_per_token == 0) mem_per_token = 1;
            return true;
        }
        logits.resize((size_t)V * (all_logits ? N : 1));
        if (args.embedding_eval_enabled) embeddings.resize((size_t)hp.n_embd);
        for (int i = 0; i < N; i += max_batch) {
            const int n = std::min(max_batch, N - i);
            float *dst = all_logits ? logits.data() + (size_t)i * V : logits.data();
            if (fl_model_eval(model, toks.data() + i, n, past + i, dst, all_logits ? 1 : 0,
                              args.embedding_eval_enabled ? embeddings.data() : nullptr) != FL_OK) {
                log.err("Model::eval", std::string(fl_last_error()) + "\n");
                return false;
            }
        }
        if (mem_per_token == 0) mem_per_token = 1;
        return true;
    }

    // The evals of one ingest() call, collected instead of run one by one: the bookkeeping between them (n_past, the context
    // recycling) never looks at an eval's results, so a run of consecutive blocks can go to the device as ONE pipelined ingest
    // (fl_model_ingest: two blocks in flight, lm-head for the last one only) -- the same evals, the same K/V cache and logits.
    struct PendingEval { int past; std::vector<token_t> toks; };
    std::vector<PendingEval> pending;
    // returns the index of the first entry that failed (all evals before it are in the K/V cache), or pending.size()
    size_t flush_pending() {
        bool ok = true;
        size_t i = 0, failed = pending.size();
        while (ok && i < pending.size()) {
            size_t j = i + 1;                                       // [i, j): consecutive positions, sizes the device accepts
            int end = pending[i].past + (int)pending[i].toks.size();
            const bool plain = !all_logits && !args.embedding_eval_enabled && (int)pending[i].toks.size() <= max_batch;
            while (plain && j < pending.size() && pending[j].past == end && (int)pending[j].toks.size() <= max_batch) {
                end += (int)pending[j].toks.size();
                ++j;
            }
            if (j - i >= 2) {
                std::vector<token_t> all;
                std::vector<int> lens;
                for (size_t k = i; k < j; ++k) {
                    all.insert(all.end(), pending[k].toks.begin(), pending[k].toks.end());
                    lens.push_back((int)pending[k].toks.size());
                }
                logits_on_device = 0;
                logits.resize((size_t)hp.n_vocab);
                if (fl_model_ingest(model, all.data(), lens.data(), (int)lens.size(), pending[i].past, logits.data()) != FL_OK) {
                    // which block failed?  The reference stops AT the failing block (lib/bridge.cpp:214-219), with every block before
                    // it in the K/V cache: take the group again one eval at a time (the K/V rows are simply rewritten)
                    log.err("Model::eval", std::string(fl_last_error()) + "\n");
                    for (size_t k = i; k < j && ok; ++k) {
                        ok = eval(pending[k].past, pending[k].toks);
                        if (!ok) failed = k;
                    }
                }
                if (mem_per_token == 0) mem_per_token = 1;
            } else {
                ok = eval(pending[i].past, pending[i].toks);
                if (!ok) failed = i;
            }
            i = j;
        }
        return failed;
    }

    // lib/bridge.cpp:161-180
    bool recycle_if_exceeds_context() {
        const size_t len = embd.size();
        if (len == 0 || (int)len + n_past <= args.n_ctx) return false;
        const size_t last_len = last_n.size();
        const size_t remaining = (size_t)(n_past - std::min(keep, n_past));
        const size_t begin_pos = last_len - std::min(remaining >> 1, last_len);
        n_past = keep;
        if (begin_pos < system_prompt.size()) {
            embd.insert(embd.begin(), system_prompt.begin(), system_prompt.end());
            return true;
        }
        embd.insert(embd.begin(), last_n.end() - (std::ptrdiff_t)begin_pos, last_n.end());
        embd.insert(embd.begin(), system_prompt.begin(), system_prompt.end());
        return true;
    }

    // lib/bridge.cpp:186-238: the block staged last is evaluated by the NEXT ingest()/generate() call
    bool ingest(std::string prompt, bool is_system) {
        log.reset();
        prompt.insert(0, 1, ' ');
        const std::vector<token_t> in = tokenize(vocab, prompt, true);
        const int max_in = args.n_ctx - 4;
        if ((int)in.size() > max_in) {
            log.err("ingest", "prompt size(='" + std::to_string(in.size()) + "') exceeds maximum allowed size('" + std::to_string(max_in) + "')");
            return false;
        }
        if (is_system) {
            if (keep < (int)in.size()) {
                log.err("ingest", "system prompt size(='" + std::to_string(in.size()) + "') exceeds 'n_keep'(='" + std::to_string(keep) + "')");
                return false;
            }
            system_prompt = in;
        }
        const size_t nb = (size_t)args.n_batch;
        pending.clear();                                            // (nothing survives a call that an exception cut short)
        for (size_t i = 0; i < in.size(); i += nb) {
            log.progress(PROGRESS_TAG_INGEST, i, in.size());
            const size_t block = std::min(nb, in.size() - i);
            recycle_if_exceeds_context();
            if (!embd.empty()) pending.push_back({n_past, embd});   // (the reference evaluates here; see flush_pending)
            n_past += (int)embd.size();
            embd.assign(in.begin() + (std::ptrdiff_t)i, in.begin() + (std::ptrdiff_t)(i + block));
            for (size_t j = 0; j < block; ++j) push_last(in[i + j]);
        }
        const size_t failed = flush_pending();
        if (failed < pending.size()) {
            // the reference returns from inside the loop with n_past and embd as they were when that eval failed
            // (lib/bridge.cpp:214-219): the K/V cache holds everything before the failing block, the block itself is still staged
            n_past = pending[failed].past;
            embd = pending[failed].toks;
            pending.clear();
            return false;
        }
        pending.clear();
        log.progress(PROGRESS_TAG_INGEST, in.size(), in.size());
        last_n.clear();
        return true;
    }

    // lib/bridge.cpp:240-312
    bool generate(const std::function<void(const std::string &)> &fn, size_t num_tokens, float top_k, float top_p, float temp,
                  float repeat_penalty, const std::vector<std::string> &stops) {
        log.reset();
        size_t max_buf = 0;
        for (const auto &w : stops) max_buf = std::max(max_buf, tokenize(vocab, w, false).size());
        TokenBuffer tb{vocab, max_buf, fn, {}, {}};
        tb.restore(tb_state);
        for (size_t i = 0; i < num_tokens; ++i) {
            std::string before, after;
            if (tb.find_stop(stops, before, after)) {
                fn(before);
                tb_state.unicode_backlog = tb.backlog;
                tb_state.left_out = after;
                return true;
            }
            recycle_if_exceeds_context();
            if (!embd.empty() && !eval(n_past, embd)) return false;
            n_past += (int)embd.size();
            embd.clear();
            sync_logits();                 // (a perplexity() call may have left its block's logits in HBM)
            if (logits.size() < (size_t)hp.n_vocab) {
                log.err("generate", "no logits to sample from: ingest a prompt first\n");
                return false;
            }
            const float *last = logits.data() + logits.size() - (size_t)hp.n_vocab;
            const token_t id = sample_top_p_top_k(last, hp.n_vocab, last_n, (double)repeat_penalty, (int)top_k, (double)top_p,
                                                  (double)temp, rng);
            if (id == TOKEN_EOS) break;
            push_last(id);
            tb.add(id);
            embd.push_back(id);
        }
        while (!tb.buf.empty()) tb.flush_one();
        return true;
    }

    // lib/bridge.cpp:331-422
    float perplexity(std::string_view prompt) {
        const bool old = all_logits;
        all_logits = true;
        const std::vector<token_t> toks = tokenize(vocab, prompt, true);
        const size_t bs = (size_t)args.n_batch, V = (size_t)hp.n_vocab;
        const size_t blocks = toks.size() / bs + (toks.size() % bs != 0);
        log.info("perplexity", "calculating perplexity over " + std::to_string(blocks) + " chunk(s)\n");
        double nll = 0.0, res = 0.0;
        size_t count = 0, idx = 1;
        for (size_t i = 0; i < toks.size(); i += bs, ++idx) {
            const size_t block = std::min(bs, toks.size() - i);
            const std::vector<token_t> in(toks.begin() + (std::ptrdiff_t)i, toks.begin() + (std::ptrdiff_t)(i + block));
            if (!eval(0, in, true)) { all_logits = old; return -1.f; }
            // softmax(logits[j])[next token] over the second half of the block (lib/bridge.cpp:397-407) -- on the device
            // (fl_model_logits_nll): one double per row comes back instead of n_batch x n_vocab floats (64 MB at 512 x 32000);
            // the rows are summed in order.  llama_get_logits() fetches the block's logits if a caller asks for them.
            const size_t j0 = block >> 1, j1 = block > 0 ? block - 1 : 0;
            if (j1 > j0) {
                std::vector<double> row_nll(j1 - j0);
                std::vector<int32_t> next(j1 - j0);
                for (size_t j = j0; j < j1; ++j) next[j - j0] = toks[i + j + 1];
                bool ok;
                if (logits_on_device > 0) {
                    ok = fl_model_logits_nll(model, (int)j0, (int)(j1 - j0), next.data(), row_nll.data()) == FL_OK;
                } else {                                   // (a block larger than the device batch was evaluated in pieces)
                    ok = true;
                    for (size_t j = j0; j < j1; ++j) {
                        const float *l = logits.data() + j * V;
                        // the reference's own arithmetic (softmax(), lib/bridge.cpp:314-330, then :405): a float sum of float
                        // exponentials taken in order, a float division, a float log -- with the host's libm, as there
                        const float mx = *std::max_element(l, l + V);
                        float sum = 0.f;
                        for (size_t k = 0; k < V; ++k) sum += std::exp(l[k] - mx);
                        const float pr = std::exp(l[(size_t)next[j - j0]] - mx) / sum;
                        row_nll[j - j0] = (double)(-std::log(pr));
                    }
                }
                if (!ok) { log.err("perplexity", std::string(fl_last_error()) + "\n"); all_logits = old; return -1.f; }
                for (double v : row_nll) nll += v;
                count += j1 - j0;
            }
            res = std::exp(nll / (double)count);
            char line[96];
            snprintf(line, sizeof line, "[%zu/%zu]: %.4f\n", idx, blocks, res);
            log.info("perplexity", line);
        }
        all_logits = old;
        return (float)res;
    }

    // session state file: byte-compatible with the reference's (lib/bridge.cpp:424-525 + KV dump lib/llama.cpp:57-78)
    bool save_state(const char *path) {
        sync_logits();
        FILE *f = fopen(path, "wb");
        if (!f) { log.err("save_state", "unable to open the file saving the model state"); return false; }
        auto W = [&](const void *p, size_t n) { return fwrite(p, 1, n, f) == n; };
        bool ok = W(&n_past, sizeof n_past);
        std::stringstream ss;
        ss << rng;
        const std::string rs = ss.str();
        size_t n = rs.size();
        ok = ok && W(&n, sizeof n) && W(rs.data(), n) && W(&mem_per_token, sizeof mem_per_token);
        n = embd.size();
        ok = ok && W(&n, sizeof n) && W(embd.data(), n * sizeof(token_t));
        n = last_n.size();
        ok = ok && W(&n, sizeof n);
        for (token_t t : last_n) ok = ok && W(&t, sizeof t);
        n = logits.size();
        ok = ok && W(&n, sizeof n) && W(logits.data(), n * sizeof(float));
        n = system_prompt.size();
        ok = ok && W(&n, sizeof n) && W(system_prompt.data(), n * sizeof(token_t));
        const int32_t memory_type = 0;   // GGML_TYPE_F32, include/llama.hpp:111
        ok = ok && W(&memory_type, sizeof memory_type);
        const size_t kv = (size_t)hp.n_layer * args.n_ctx * hp.n_embd;
        std::vector<float> k(kv), v(kv);
        ok = ok && fl_model_kv_read(model, k.data(), v.data()) == FL_OK && W(k.data(), kv * 4) && W(v.data(), kv * 4);
        fclose(f);
        if (!ok) log.err("save_state", "failed to write the model state\n");
        return ok;
    }

    bool load_state(const char *path) {
        // Every size field is checked against what is left of the file and against the model's own limits, everything is
        // read into temporaries, and the session is touched only once the whole file (KV cache included) has been read.
        FILE *f = fopen(path, "rb");
        if (!f) { log.err("load_state", "unable to open the file loading the model state"); return false; }
        fseek(f, 0, SEEK_END);
        const long fsize = ftell(f);
        fseek(f, 0, SEEK_SET);
        auto left = [&]() -> size_t { const long p = ftell(f); return p < 0 || p > fsize ? 0 : (size_t)(fsize - p); };
        auto R = [&](void *p, size_t n) { return n <= left() && fread(p, 1, n, f) == n; };
        auto Rn = [&](size_t *n, size_t elem, size_t limit) { return R(n, sizeof *n) && *n <= limit && *n * elem <= left(); };
        int t_past = 0;
        size_t t_mem = 0, n = 0;
        std::string rs;
        std::vector<token_t> t_embd, t_last, t_sys;
        std::vector<float> t_logits;
        const size_t ctx = (size_t)args.n_ctx, V = (size_t)hp.n_vocab;
        bool ok = R(&t_past, sizeof t_past) && t_past >= 0 && (size_t)t_past <= ctx;
        ok = ok && Rn(&n, 1, 1 << 20);
        if (ok) { rs.resize(n); ok = R(rs.data(), n); }
        std::mt19937 t_rng;
        if (ok) { std::stringstream ss; ss << rs; ss >> t_rng; ok = !ss.fail(); }
        ok = ok && R(&t_mem, sizeof t_mem) && Rn(&n, sizeof(token_t), ctx + (size_t)max_batch);
        if (ok) { t_embd.resize(n); ok = R(t_embd.data(), n * sizeof(token_t)); }
        ok = ok && Rn(&n, sizeof(token_t), 1 << 20);
        if (ok) { t_last.resize(n); ok = R(t_last.data(), n * sizeof(token_t)); }
        ok = ok && Rn(&n, sizeof(float), V * std::max<size_t>(ctx, (size_t)max_batch));
        if (ok) { t_logits.resize(n); ok = R(t_logits.data(), n * sizeof(float)); }
        ok = ok && Rn(&n, sizeof(token_t), ctx);
        if (ok) { t_sys.resize(n); ok = R(t_sys.data(), n * sizeof(token_t)); }
        int32_t memory_type = 0;
        ok = ok && R(&memory_type, sizeof memory_type) && memory_type == 0;
        const size_t kv = (size_t)hp.n_layer * args.n_ctx * hp.n_embd;
        std::vector<float> k, v;
        if (ok && 2 * kv * 4 <= left()) { k.resize(kv); v.resize(kv); } else ok = false;
        ok = ok && R(k.data(), kv * 4) && R(v.data(), kv * 4);
        fclose(f);
        for (token_t t : t_embd) ok = ok && t >= 0 && (size_t)t < V;
        ok = ok && fl_model_kv_write(model, k.data(), v.data()) == FL_OK;
        if (!ok) { log.err("load_state", "failed to read the model state (was it saved with the same model and n_ctx?)\n"); return false; }
        n_past = t_past;
        rng = t_rng;
        mem_per_token = t_mem;
        embd = std::move(t_embd);
        last_n.clear();
        for (token_t t : t_last) push_last(t);
        logits = std::move(t_logits);
        logits_on_device = 0;
        system_prompt = std::move(t_sys);
        return true;
    }

    bool reset() {
        log.info("reset", "resetting the model...\n");
        n_past = 0;
        last_n.clear();
        logits.clear();
        logits_on_device = 0;
        system_prompt.clear();
        embd.clear();
        rng = std::mt19937((uint32_t)seed);
        log.info("reset", "reset completed.\n");
        return true;
    }
};

}  // namespace

struct llama_model_context {
    llama_model_context_args args{};
    std::unique_ptr<Session> inner;
    std::vector<std::string> stop_words;
};

static bool valid(const llama_model_context *c) {
    if (!c) { fprintf(stderr, "model context is not initalized. Please use `llama_create_context` to create a context.\n"); return false; }
    if (!c->inner) { fprintf(stderr, "model is not loaded. Please use `llama_load_model` to load a model.\n"); return false; }
    return true;
}

// No exception crosses the C ABI (the reference is built -fno-exceptions and returns false; SURVEY.md 8b "Errors"):
// a std::bad_alloc / length_error from a corrupt file or an out-of-memory host becomes `false` (-1 for perplexity).
template <class F, class R>
static R guarded(R fail, F &&f) noexcept {
    try {
        return f();
    } catch (const std::exception &e) {
        fprintf(stderr, "fastllama_hip: %s\n", e.what());
    } catch (...) {
        fprintf(stderr, "fastllama_hip: unknown exception\n");
    }
    return fail;
}

extern "C" {

struct llama_model_context_args llama_create_default_context_args(void) {
    llama_model_context_args a{};                       // FastLlama::Params defaults, include/bridge.hpp:21-36
    a.embedding_eval_enabled = false;
    a.should_get_all_logits = false;
    a.use_mmap = false;
    a.use_mlock = false;
    a.load_parallel = false;
    a.seed = 0;
    a.n_keep = 64;
    a.n_ctx = 512;
    a.n_threads = 1;
    a.n_batch = 16;
    a.n_load_parallel_blocks = 1;
    a.last_n_tokens = 64;
    a.allocate_extra_mem = 0;
    a.logger = {def_log_info, def_log_err, def_log_warn, def_log_reset, def_log_progress};
    return a;
}

struct llama_model_context *llama_create_context(struct llama_model_context_args args) {
    auto *c = new (std::nothrow) llama_model_context();
    if (c) c->args = args;
    return c;
}

bool llama_load_model(struct llama_model_context *c, char const *filepath) {
    if (!c) { fprintf(stderr, "model context is not initalized. Please use `llama_create_context` to create a context.\n"); return false; }
    if (c->inner) { fprintf(stderr, "model is already loaded.\n"); return false; }
    if (!filepath) return false;
    return guarded(false, [&]() -> bool {
    auto s = std::make_unique<Session>();
    s->args = c->args;
    s->log.cb = c->args.logger;
    g_warn_log.cb = c->args.logger;
    fl_set_warn_handler(forward_warning);
    s->seed = c->args.seed;
    s->keep = c->args.n_keep;
    s->rng = std::mt19937((uint32_t)c->args.seed);
    s->last_n_cap = c->args.last_n_tokens;
    s->last_n.assign(c->args.last_n_tokens, 0);      // RingBuffer(size) starts with `size` zero tokens, ring_buffer.hpp:26-29
    s->all_logits = c->args.should_get_all_logits;
    if (c->args.n_ctx <= 8 || c->args.n_batch <= 0) { s->log.err("FastLlama::Params::build", "bad n_ctx / n_batch\n"); return false; }
    if (c->args.n_ctx % 4 != 0) s->args.n_ctx = c->args.n_ctx = (c->args.n_ctx + 3) / 4 * 4;
    if (!s->load(filepath)) {
        s->log.err("FastLlama::Params::build", "Unable to load model\n");
        return false;
    }
    c->inner = std::move(s);
    return true;
    });
}

bool llama_set_stop_words(struct llama_model_context *c, char const **words, size_t len) {
    if (!c) { fprintf(stderr, "model context is not initalized. Please use `llama_create_context` to create a context.\n"); return false; }
    c->stop_words.assign(len, std::string());
    for (size_t i = 0; i < len; ++i) c->stop_words[i] = words[i] ? words[i] : "";
    return true;
}

bool llama_ingest(struct llama_model_context *c, char const *prompt) {
    return guarded(false, [&] { return valid(c) && prompt && c->inner->ingest(prompt, false); });
}
bool llama_ingest_system_prompt(struct llama_model_context *c, char const *prompt) {
    return guarded(false, [&] { return valid(c) && prompt && c->inner->ingest(prompt, true); });
}

bool llama_generate(struct llama_model_context *c, LLAMA_STREAM_FUNC stream_fn, size_t number_of_tokens, float top_k, float top_p,
                    float temp, float repeat_penalty) {
    if (!valid(c)) return false;
    return guarded(false, [&] {
        return c->inner->generate([stream_fn](const std::string &s) { if (stream_fn) stream_fn(s.data(), (int)s.size()); },
                                  number_of_tokens, top_k, top_p, temp, repeat_penalty, c->stop_words);
    });
}

float llama_perplexity(struct llama_model_context *c, char const *prompt) {
    if (!valid(c) || !prompt) return -1.f;
    return guarded(-1.f, [&] { return c->inner->perplexity(prompt); });
}

struct llama_array_view_f llama_get_embeddings(struct llama_model_context const *c) {
    if (!valid(c)) return {nullptr, 0};
    if (!c->inner->args.embedding_eval_enabled)
        c->inner->log.warn("get_embeddings", "Please set the flag `embeddings_eval_enable` to true before getting the embeddings.\n");
    return {c->inner->embeddings.data(), c->inner->embeddings.size()};
}

struct llama_array_view_f llama_get_logits(struct llama_model_context const *c) {
    if (!valid(c)) return {nullptr, 0};
    guarded(0, [&] { c->inner->sync_logits(); return 0; });
    return {c->inner->logits.data(), c->inner->logits.size()};
}

bool llama_save_state(struct llama_model_context *c, char const *path) {
    return guarded(false, [&] { return valid(c) && path && c->inner->save_state(path); });
}
bool llama_load_state(struct llama_model_context *c, char const *path) {
    return guarded(false, [&] { return valid(c) && path && c->inner->load_state(path); });
}

bool llama_attach_lora(struct llama_model_context *c, char const *path) {
    return guarded(false, [&] { return valid(c) && path && c->inner->attach_lora(path); });
}
bool llama_detach_lora(struct llama_model_context *c) { return guarded(false, [&] { return valid(c) && c->inner->detach_lora(); }); }

bool llama_reset_model(struct llama_model_context *c) { return guarded(false, [&] { return valid(c) && c->inner->reset(); }); }
void llama_free_context(struct llama_model_context *c) { delete c; }

void llama_handle_signal(int) {
    printf("Quitting the app...");
    exit(0);
}

}  // extern "C"
