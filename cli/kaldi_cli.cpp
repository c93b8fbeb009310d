// cli/kaldi_cli.cpp -- drop-in command-line tools kaldi_spectrogram, kaldi_fbank, kaldi_mfcc, kaldi_plp.
//
// Same flags, defaults, validation messages and exit codes as the reference's generated CLIs (srci/kaldi_mfcc.cpp:84-172
// and src/kaldi_mfcc.cpp:118-327; fbank srci/kaldi_fbank.cpp:89-106; spectrogram src/kaldi_spectrogram.cpp; plp
// srci/kaldi_plp.cpp:84-101), but linked against libspeech.so (include/codee_speech.h) instead of unity-including the
// .c file.  One source, built four times with -DTOOL=<n> (cli/Makefile).
//
// Differences, both forced by what is absent here: argtable2 is replaced by getopt_long (same short and long names), and
// the tensor header of the reference's sibling repo (util/cmli.hpp, not in the tree) is replaced by the public .npy
// format: the input is a 1-D (or 1xN / Nx1) '<f4' or '<f8' .npy, the output a 2-D .npy of the same dtype, W x D for a
// C-order input and D x W (fortran_order) for a Fortran-order input -- "samples from one frame are contiguous in memory"
// either way (src/kaldi_mfcc.cpp:78-81, :259-260).  kaldi_fbank accepts float only, as the reference (srci/kaldi_fbank.cpp:6).
//
// Widening: --ark=<utt-id> writes the features as one entry of a Kaldi binary archive instead of a .npy
// ("<utt-id> \0B" "FM " \4 <int32 rows> \4 <int32 cols> <rows x cols floats>, "DM " for double -- the layout the
// reference's kaldi2bin scans for, src/kaldi2bin.cpp:141-175), --append adds it to an existing archive, and
// --scp=<file> appends the matching "<utt-id> <ark>:<offset>" line, so the tools can stand in for compute-mfcc-feats.
// Widening: an input that starts with "RIFF" is read as a 16-bit PCM WAVE file (--channel=<n> picks the
// channel, default 0); the samples go to the device as int16 (speech_extract_pcm16, include/speech_hip.h) and keep their
// integer scale, as Kaldi's wave reader leaves them.  The sample rate comes from the header unless -r is given, and an
// -r that contradicts the header is an error.  The output is float32.
// --raw=<f32|f64|s16> reads the input as headerless samples of that type up to the end of the file or stream and writes
// the features headerless too (f32 for s16 input), frame-major -- the raw tensor I/O of SURVEY.md 8(f1).
// --wav_scp=<file> takes a Kaldi wav.scp ("<utt-id> <path to .wav>" per line; no pipes) instead of one input file and writes
// every utterance's features into one archive (-o, with --scp the matching script file): the whole list goes through
// speech_plan_run_host in groups, int16 over PCIe, the usage of compute-mfcc-feats scp:wav.scp ark,scp:feats.ark,feats.scp.
#include <cctype>
#include <cfloat>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <getopt.h>
#include <iostream>
#include <string>
#include <unistd.h>
#include <utility>
#include <vector>
#include "../include/codee_speech.h"
#include "../include/speech_hip.h"

#define TOOL_SPECTROGRAM 0
#define TOOL_FBANK 1
#define TOOL_MFCC 2
#define TOOL_PLP 3
#ifndef TOOL
#define TOOL TOOL_MFCC
#endif

using namespace std;

static const char *prog()
{
    return TOOL == TOOL_SPECTROGRAM ? "kaldi_spectrogram" : TOOL == TOOL_FBANK ? "kaldi_fbank" : TOOL == TOOL_MFCC ? "kaldi_mfcc" : "kaldi_plp";
}

static int fail(int line, const string &msg)
{
    cerr << prog() << ": " << line << ": \033[1;31merror:\033[0m " << msg << endl;       // src/kaldi_mfcc.cpp:32
    return 1;
}
#define FAIL(msg) return fail(__LINE__, msg)

struct NpyInfo { bool f64 = false, fortran = false; vector<size_t> shape; size_t n = 0; };

// .npy v1/v2/v3 header (public format): magic, version, little-endian header length, python dict literal
static bool read_npy_header(istream &is, NpyInfo &h, const char magic[8])
{
    if (memcmp(magic, "\x93NUMPY", 6) != 0) { return false; }
    size_t hlen = 0;
    if (magic[6] == 1) { unsigned char b[2]; if (!is.read((char *)b, 2)) { return false; } hlen = b[0] | (b[1] << 8); }
    else { unsigned char b[4]; if (!is.read((char *)b, 4)) { return false; } hlen = b[0] | (b[1] << 8) | (b[2] << 16) | ((size_t)b[3] << 24); }
    if (hlen > (1u << 20)) { return false; }
    string d(hlen, '\0');
    if (!is.read(&d[0], (streamsize)hlen)) { return false; }
    const size_t pd = d.find("'descr'"), pf = d.find("'fortran_order'"), ps = d.find("'shape'");
    if (pd == string::npos || pf == string::npos || ps == string::npos) { return false; }
    const size_t q1 = d.find('\'', d.find(':', pd)), q2 = d.find('\'', q1 + 1);
    if (q1 == string::npos || q2 == string::npos) { return false; }
    const string descr = d.substr(q1 + 1, q2 - q1 - 1);
    if (descr == "<f4" || descr == "|f4" || descr == "=f4") { h.f64 = false; }
    else if (descr == "<f8" || descr == "|f8" || descr == "=f8") { h.f64 = true; }
    else { return false; }
    h.fortran = d.compare(d.find_first_not_of(" ", d.find(':', pf) + 1), 4, "True") == 0;
    const size_t p1 = d.find('(', ps), p2 = d.find(')', p1);
    if (p1 == string::npos || p2 == string::npos) { return false; }
    h.shape.clear(); h.n = 1;
    const string sh = d.substr(p1 + 1, p2 - p1 - 1);
    size_t i = 0;
    while (i < sh.size())
    {
        while (i < sh.size() && !isdigit((unsigned char)sh[i])) { ++i; }
        if (i >= sh.size()) { break; }
        size_t v = 0;
        while (i < sh.size() && isdigit((unsigned char)sh[i])) { v = v * 10 + (size_t)(sh[i] - '0'); ++i; }
        h.shape.push_back(v); h.n *= v;
    }
    return true;
}

// RIFF/WAVE, 16-bit PCM (public format).  `magic` holds the first 8 bytes ("RIFF" + size).  Reads up to the end of the data
// chunk (or of the stream when the chunk size is 0 or 0xFFFFFFFF, as streamed files have it) and returns one channel.
struct WavInfo { unsigned channels = 0, rate = 0; vector<short> x; };
static bool read_wav(istream &is, const char magic[8], unsigned channel, WavInfo &w, string &why)
{
    auto u16 = [](const unsigned char *b) { return (unsigned)(b[0] | (b[1] << 8)); };
    auto u32 = [](const unsigned char *b) { return (unsigned)b[0] | ((unsigned)b[1] << 8) | ((unsigned)b[2] << 16) | ((unsigned)b[3] << 24); };
    unsigned char b[40];
    if (memcmp(magic, "RIFF", 4) != 0 || !is.read((char *)b, 4) || memcmp(b, "WAVE", 4) != 0) { why = "not a RIFF/WAVE file"; return false; }
    bool have_fmt = false;
    for (;;)
    {
        if (!is.read((char *)b, 8)) { why = "no data chunk"; return false; }
        const unsigned size = u32(b + 4);
        if (memcmp(b, "fmt ", 4) == 0)
        {
            if (size < 16 || size > 40 || !is.read((char *)b, size)) { why = "bad fmt chunk"; return false; }
            unsigned tag = u16(b);
            if (tag == 0xFFFEu && size >= 26) { tag = u16(b + 24); }                         // WAVE_FORMAT_EXTENSIBLE: first two bytes of the sub-format
            w.channels = u16(b + 2); w.rate = u32(b + 4);
            if (tag != 1u || u16(b + 14) != 16u) { why = "only 16-bit PCM is supported"; return false; }
            if (w.channels < 1u || u16(b + 12) != 2u * w.channels) { why = "bad channel count or block size"; return false; }
            if (size & 1u) { is.ignore(1); }
            have_fmt = true;
        }
        else if (memcmp(b, "data", 4) == 0)
        {
            if (!have_fmt) { why = "data chunk before fmt chunk"; return false; }
            if (channel >= w.channels) { why = "channel not in file"; return false; }
            const bool open_ended = (size == 0u || size == 0xFFFFFFFFu);
            const size_t frame = 2u * w.channels;
            vector<unsigned char> buf;
            if (open_ended) { buf.assign(istreambuf_iterator<char>(is), istreambuf_iterator<char>()); }
            else { buf.resize(size); is.read((char *)buf.data(), (streamsize)size); buf.resize((size_t)is.gcount()); }     // a truncated file gives what it holds
            const size_t n = buf.size() / frame;
            w.x.resize(n);
            for (size_t i = 0; i < n; ++i) { w.x[i] = (short)u16(buf.data() + i * frame + 2u * channel); }
            return true;
        }
        else { is.ignore((streamsize)size + (size & 1u)); if (!is) { why = "no data chunk"; return false; } }
    }
}

static void put_ark_header(ostream &os, const string &key, bool f64, size_t W, size_t D)
{
    // Kaldi binary matrix entry, always frame-major (W x D)
    const int32_t rows = (int32_t)W, cols = (int32_t)D;
    os << key << ' ';
    os.put('\0'); os.put('B'); os.write(f64 ? "DM " : "FM ", 3);
    os.put('\4'); os.write((const char *)&rows, 4); os.put('\4'); os.write((const char *)&cols, 4);
}

// --wav_scp: every line's .wav through the plan, in groups of at most `group_samples` samples, into one archive
static int run_wav_scp(const string &list, const speech_opts &o_in, bool have_r, unsigned channel, ostream &os, long long ark_off, const string &ofile,
                       const string &scp_file, size_t group_samples, const vector<int> &devices)
{
    ifstream ls(list);
    if (!ls) { FAIL("problem opening the wav_scp file"); }
    struct Item { string key; WavInfo wav; };
    vector<pair<string, string>> entries;
    string line;
    while (getline(ls, line))
    {
        const size_t a = line.find_first_not_of(" \t");
        if (a == string::npos) { continue; }
        const size_t b = line.find_first_of(" \t", a);
        if (b == string::npos) { FAIL("wav_scp line without a path: " + line); }
        const size_t c = line.find_first_not_of(" \t", b), e = line.find_last_not_of(" \t\r");
        if (c == string::npos) { FAIL("wav_scp line without a path: " + line); }
        const string path = line.substr(c, e - c + 1);
        if (!path.empty() && path.back() == '|') { FAIL("wav_scp pipes are not supported: " + line); }
        entries.emplace_back(line.substr(a, b - a), path);
    }
    if (entries.empty()) { FAIL("the wav_scp file lists no utterances"); }
    ofstream scp;
    if (!scp_file.empty()) { scp.open(scp_file, ios::app); if (!scp) { FAIL("problem opening scp file"); } }
    speech_opts o = o_in;
    speech_plan *plan = nullptr;
    size_t D = 0, done = 0;
    while (done < entries.size())
    {
        vector<Item> items;
        size_t total = 0;
        while (done + items.size() < entries.size() && (items.empty() || total < group_samples))
        {
            const auto &en = entries[done + items.size()];
            ifstream f(en.second, ios::binary);
            if (!f) { FAIL("problem opening input file " + en.second); }
            Item it; it.key = en.first;
            char magic[8]; string why;
            if (!f.read(magic, 8) || !read_wav(f, magic, channel, it.wav, why)) { FAIL("problem reading .wav input " + en.second + ": " + why); }
            if (it.wav.x.empty()) { FAIL("input (X) found to be empty: " + en.second); }
            if (!plan)
            {
                if (have_r && o.sr != (double)it.wav.rate) { FAIL("srate (sample rate) does not match the .wav header (" + to_string(it.wav.rate) + " Hz): " + en.second); }
                o.sr = (double)it.wav.rate;
                if (o.kind != SPEECH_SPECTROGRAM && o_in.hif < DBL_EPSILON) { o.hif = 0.5 * o.sr + o_in.hif; }
                if (o.kind != SPEECH_SPECTROGRAM && o.hif <= o.lof) { FAIL("hif must be > lof"); }
                if (speech_plan_create(&plan, &o, devices.empty() ? 0 : devices[0])) { FAIL("problem during function call"); }
                speech_plan_sizes(plan, nullptr, nullptr, nullptr, &D, nullptr);
            }
            else if ((double)it.wav.rate != o.sr) { speech_plan_destroy(plan); FAIL("sample rate differs from the first file's: " + en.second); }
            total += it.wav.x.size();
            items.push_back(std::move(it));
        }
        const size_t n = items.size();
        vector<const void *> xp(n); vector<void *> yp(n); vector<size_t> nn(n), ww(n); vector<unsigned> ids(n);
        vector<vector<float>> ys(n);
        for (size_t i = 0; i < n; ++i)
        {
            nn[i] = items[i].wav.x.size();
            if (speech_plan_num_frames(plan, nn[i], &ww[i])) { speech_plan_destroy(plan); FAIL("frame length must be < signal length if snip_edges: " + items[i].key); }
            ys[i].resize(ww[i] * D + 1);
            xp[i] = items[i].wav.x.data(); yp[i] = ys[i].data(); ids[i] = (unsigned)(done + i);
        }
        // --devices: the group is dealt over the listed GPUs inside the library (speech_run_host_multi: a thread, a plan and a
        // staging pipeline per device, no collective); the archive is the same to the bit, dither included
        const int rc = (devices.size() > 1) ? speech_run_host_multi(&o, (int)devices.size(), devices.data(), n, xp.data(), nn.data(), yp.data(), 1, 0, ids.data())
                                            : speech_plan_run_host(plan, n, xp.data(), nn.data(), yp.data(), 1, 0, ids.data());
        if (rc) { speech_plan_destroy(plan); FAIL("problem during function call"); }
        for (size_t i = 0; i < n; ++i)
        {
            put_ark_header(os, items[i].key, false, ww[i], D);
            os.write((const char *)ys[i].data(), (streamsize)(ww[i] * D * sizeof(float)));
            if (!os) { speech_plan_destroy(plan); FAIL("problem writing output file (Y)"); }
            if (scp.is_open()) { scp << items[i].key << ' ' << ofile << ':' << (ark_off + (long long)items[i].key.size() + 1) << '\n'; }
            ark_off += (long long)items[i].key.size() + 1 + 15 + (long long)(ww[i] * D * sizeof(float));
        }
        done += n;
    }
    os.flush();
    speech_plan_destroy(plan);
    if (scp.is_open() && !scp) { FAIL("problem writing scp file"); }
    return 0;
}

static bool write_npy_header(ostream &os, bool f64, bool fortran, size_t r, size_t c)
{
    string d = string("{'descr': '") + (f64 ? "<f8" : "<f4") + "', 'fortran_order': " + (fortran ? "True" : "False") + ", 'shape': (" + to_string(r) + ", " + to_string(c) + "), }";
    while ((10 + d.size() + 1) % 64 != 0) { d += ' '; }
    d += '\n';
    const unsigned char pre[10] = {0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, (unsigned char)(d.size() & 0xFF), (unsigned char)(d.size() >> 8)};
    os.write((const char *)pre, 10); os.write(d.data(), (streamsize)d.size());
    return (bool)os;
}

static void usage()
{
    cout << "Usage: " << prog() << " [<file>] [-r <dbl>] [-t <dbl>] [-s <dbl>] [-e] [-d <dbl>] [-z] [-y] [-p <dbl>] [-w <str>]";
    if (TOOL == TOOL_FBANK) { cout << " [-a] [-l <dbl>] [-u <dbl>] [-b <uint>] [-g] [-x]"; }
    if (TOOL == TOOL_MFCC || TOOL == TOOL_PLP) { cout << " [-l <dbl>] [-u <dbl>] [-b <uint>] [-c <uint>] [-q <dbl>] [-x]"; }
    if (TOOL == TOOL_PLP) { cout << " [--compress=<dbl>] [--cep_scale=<dbl>]"; }
    cout << " [-m] [-o <file>] [-h]\n\n";
    const char *fmt = "  %-25s %s\n";
    printf(fmt, "<file>", "input file (X), .npy float32/float64 vector or 16-bit PCM .wav; stdin if absent or -");
    printf(fmt, "-r, --srate=<dbl>", "sample rate in Hz [default=16000.0]");
    printf(fmt, "-t, --frame_length=<dbl>", "length in ms of each frame [default=25]");
    printf(fmt, "-s, --frame_shift=<dbl>", "step in ms between each frame [default=10]");
    printf(fmt, "-e, --snip_edges", "include to snip edges [default=false]");
    printf(fmt, "-d, --dither=<dbl>", "dither weight [default=0.1]");
    printf(fmt, "-z, --zero_dc", "include to zero the mean of each frame [default=false]");
    printf(fmt, "-y, --raw_energy", "include to compute energy before preemph [default=false]");
    printf(fmt, "-p, --preemph=<dbl>", "preemphasis coefficient in [0 1] [default=0.97]");
    printf(fmt, "-w, --wintype=<str>", "window type [default='povey']");
    if (TOOL == TOOL_FBANK) { printf(fmt, "-a, --amp", "include to use amplitude (sqrt of power) [default=false]"); }
    if (TOOL != TOOL_SPECTROGRAM)
    {
        printf(fmt, "-l, --lof=<dbl>", "mel-bank low-freq in Hz [default=20.0]");
        printf(fmt, "-u, --hif=<dbl>", "mel-bank high-freq in Hz [default=Nyquist]; <= 0 is an offset from Nyquist");
        printf(fmt, "-b, --B=<uint>", "number of mel bins [default=23]");
    }
    if (TOOL == TOOL_FBANK) { printf(fmt, "-g, --log", "include to output log of amplitude or power [default=false]"); }
    if (TOOL == TOOL_MFCC || TOOL == TOOL_PLP)
    {
        printf(fmt, "-c, --C=<uint>", "number of cepstral coeffs [default=13]");
        printf(fmt, "-q, --Q=<dbl>", "lifter coeff [default=22.0]");
    }
    if (TOOL == TOOL_PLP)
    {
        printf(fmt, "--compress=<dbl>", "compression exponent [default=0.33333]");
        printf(fmt, "--cep_scale=<dbl>", "cepstral scale [default=1.0]");
    }
    if (TOOL != TOOL_SPECTROGRAM) { printf(fmt, "-x, --use_energy", "include to use raw energy as (first / extra) feat [default=false]"); }
    printf(fmt, "-m, --zero_mean", "include to zero the means of each feat in Y [default=false]");
    printf(fmt, "-o, --ofile=<file>", "output file (Y), .npy; stdout if - or piped");
    printf(fmt, "--ark=<utt-id>", "write Y as a Kaldi binary archive entry with this key instead of .npy");
    printf(fmt, "--append", "with --ark: append to the archive instead of replacing it");
    printf(fmt, "--scp=<file>", "with --ark and -o: append \"<utt-id> <ofile>:<offset>\" to this script file");
    printf(fmt, "--channel=<n>", "with a .wav input: the channel to use [default=0]");
    printf(fmt, "--raw=<f32|f64|s16>", "the input is headerless samples of this type; the output is headerless too");
    printf(fmt, "--wav_scp=<file>", "list of \"<utt-id> <.wav path>\" lines instead of <file>; -o is a Kaldi archive");
    printf(fmt, "--devices=<a,b,..>", "with --wav_scp: the GPUs to share the list (one pipeline each) [default: 0]");
    printf(fmt, "--seed=<n>", "dither seed [default: from the clock, as the reference: a new noise draw on every run]");
    printf(fmt, "--utt=<n>", "with --seed: the utterance number the dither generator is keyed on [default: a per-process call counter]");
    printf(fmt, "-h, --help", "display this help and exit");
    cout << "\nMakes Kaldi " << (TOOL == TOOL_SPECTROGRAM ? "spectrogram" : TOOL == TOOL_FBANK ? "mel-fbank" : TOOL == TOOL_MFCC ? "MFCC" : "PLP")
         << " features for univariate, real-valued X on an AMD MI355X (libspeech.so).\n"
         << "The number of output frames (W) is set as in Kaldi:\n"
         << "If snip_edges=true:  W = 1u + (N-L)/stp   \nIf snip_edges=false: W = (N+stp/2u) / stp \n"
         << "Samples from one frame are contiguous in memory: Y is W x D for a C-order input, D x W for a Fortran-order input.\n"
         << "Use -d0 to turn off dithering (the default 0.1 adds seeded Gaussian noise).\n\n"
         << "Examples:\n$ " << prog() << " -e -z -d0 X.npy -o Y.npy \n$ cat X.npy | " << prog() << " -e -z - > Y.npy \n";
}

int main(int argc, char *argv[])
{
    double sr = 16000.0, fl = 25.0, shft = 10.0, d = 0.1, p = 0.97, lof = 20.0, hif = 0.0, Q = 22.0, compress = 1.0 / 3.0, cep_scale = 1.0;
    long Bi = 23, Ci = 13, channel = 0, utt = -1;
    unsigned long long seed = 0;
    bool have_seed = false;
    int snipe = 0, dc0 = 0, rawe = 0, amp = 0, lg = 0, usee = 0, mn0 = 0, help = 0;
    string wintype = "povey", ofile, ark_key, scp_file, wav_scp, raw_type;
    vector<int> devices;
    bool have_o = false, append = false, have_r = false;

    static const option longopts[] = {
        {"srate", required_argument, nullptr, 'r'}, {"frame_length", required_argument, nullptr, 't'}, {"frame_shift", required_argument, nullptr, 's'},
        {"snip_edges", no_argument, nullptr, 'e'}, {"dither", required_argument, nullptr, 'd'}, {"zero_dc", no_argument, nullptr, 'z'},
        {"raw_energy", no_argument, nullptr, 'y'}, {"preemph", required_argument, nullptr, 'p'}, {"wintype", required_argument, nullptr, 'w'},
        {"amp", no_argument, nullptr, 'a'}, {"lof", required_argument, nullptr, 'l'}, {"hif", required_argument, nullptr, 'u'},
        {"B", required_argument, nullptr, 'b'}, {"log", no_argument, nullptr, 'g'}, {"C", required_argument, nullptr, 'c'}, {"Q", required_argument, nullptr, 'q'},
        {"use_energy", no_argument, nullptr, 'x'}, {"zero_mean", no_argument, nullptr, 'm'}, {"ofile", required_argument, nullptr, 'o'},
        {"help", no_argument, nullptr, 'h'}, {"compress", required_argument, nullptr, 1000}, {"cep_scale", required_argument, nullptr, 1001},
        {"ark", required_argument, nullptr, 1002}, {"scp", required_argument, nullptr, 1003}, {"append", no_argument, nullptr, 1004}, {"channel", required_argument, nullptr, 1005}, {"wav_scp", required_argument, nullptr, 1006}, {"raw", required_argument, nullptr, 1007}, {"seed", required_argument, nullptr, 1008}, {"utt", required_argument, nullptr, 1009}, {"devices", required_argument, nullptr, 1010}, {nullptr, 0, nullptr, 0}};
    const char *shortopts = TOOL == TOOL_SPECTROGRAM ? "r:t:s:ed:zyp:w:mo:h" : TOOL == TOOL_FBANK ? "r:t:s:ed:zyp:w:al:u:b:gxmo:h" : "r:t:s:ed:zyp:w:l:u:b:c:q:xmo:h";
    int c, nerrs = 0;
    opterr = 0;
    while ((c = getopt_long(argc, argv, shortopts, longopts, nullptr)) != -1)
    {
        char *end = nullptr;
        auto num = [&](double &v) { v = strtod(optarg, &end); if (end == optarg || *end) { ++nerrs; cerr << prog() << ": invalid argument \"" << optarg << "\" to option" << endl; } };
        switch (c)
        {
            case 'r': num(sr); have_r = true; break; case 't': num(fl); break; case 's': num(shft); break; case 'e': snipe = 1; break;
            case 'd': num(d); break; case 'z': dc0 = 1; break; case 'y': rawe = 1; break; case 'p': num(p); break;
            case 'w': wintype = optarg; break; case 'a': amp = 1; break; case 'l': num(lof); break; case 'u': num(hif); break;
            case 'b': Bi = strtol(optarg, &end, 10); if (end == optarg || *end) { ++nerrs; } break;
            case 'g': lg = 1; break;
            case 'c': Ci = strtol(optarg, &end, 10); if (end == optarg || *end) { ++nerrs; } break;
            case 'q': num(Q); break; case 'x': usee = 1; break; case 'm': mn0 = 1; break;
            case 'o': ofile = optarg; have_o = true; break; case 'h': help = 1; break;
            case 1000: if (TOOL == TOOL_PLP) { num(compress); } else { ++nerrs; } break;
            case 1001: if (TOOL == TOOL_PLP) { num(cep_scale); } else { ++nerrs; } break;
            case 1002: ark_key = optarg; break;
            case 1003: scp_file = optarg; break;
            case 1004: append = true; break;
            case 1006: wav_scp = optarg; break;
            case 1007: raw_type = optarg; if (raw_type != "f32" && raw_type != "f64" && raw_type != "s16") { ++nerrs; } break;
            case 1008: seed = strtoull(optarg, &end, 0); have_seed = true; if (end == optarg || *end) { ++nerrs; } break;
            case 1009: utt = strtol(optarg, &end, 10); if (end == optarg || *end || utt < 0) { ++nerrs; } break;
            case 1010:
            {
                const char *q = optarg;
                while (*q) { const long v = strtol(q, &end, 10); if (end == q || v < 0 || v > 1023) { ++nerrs; break; } devices.push_back((int)v); q = (*end == ',') ? end + 1 : end; if (*end && *end != ',') { ++nerrs; break; } }
                if (devices.empty()) { ++nerrs; }
                break;
            }
            case 1005: channel = strtol(optarg, &end, 10); if (end == optarg || *end || channel < 0) { ++nerrs; } break;
            default: ++nerrs; cerr << prog() << ": invalid option \"" << argv[optind - 1] << "\"" << endl; break;
        }
    }
    if (help) { usage(); return 1; }                                                        // src/kaldi_mfcc.cpp:142-147: help exits with 1
    if (nerrs > 0) { FAIL("invalid command line; try -h"); }
    if (argc - optind > 1) { FAIL("too many input files"); }
    const string ifile = (optind < argc) ? argv[optind] : "";

    const bool list_mode = !wav_scp.empty();
    if (list_mode && !ifile.empty()) { FAIL("--wav_scp replaces the input file"); }
    if (!devices.empty() && !list_mode) { FAIL("--devices goes with --wav_scp"); }
    // stdin / stdout detection: src/kaldi_mfcc.cpp:151-159
    const bool stdi1 = !list_mode && (ifile.empty() || ifile == "-");
    if (stdi1 && isatty(fileno(stdin))) { FAIL("no stdin detected"); }
    const bool stdo1 = have_o ? (ofile.empty() || ofile == "-") : !isatty(fileno(stdout));
    const bool wo1 = stdo1 || have_o;

    ifstream ifs1;
    if (!stdi1 && !list_mode) { ifs1.open(ifile, ios::binary); if (!ifs1) { FAIL("problem opening input file"); } }
    istream &is = stdi1 ? cin : ifs1;
    NpyInfo i1;
    WavInfo wav;
    char magic[8] = {0};
    const bool raw_mode = !raw_type.empty();
    if (raw_mode && list_mode) { FAIL("--raw and --wav_scp exclude each other"); }
    vector<char> rawbuf;
    if (raw_mode) { rawbuf.assign(istreambuf_iterator<char>(is), istreambuf_iterator<char>()); }
    else if (!list_mode && !is.read(magic, 8)) { FAIL("problem reading header for input file"); }
    bool is_wav = memcmp(magic, "RIFF", 4) == 0;
    if (list_mode) { }
    else if (raw_mode)
    {
        const size_t es = raw_type == "f64" ? 8u : (raw_type == "f32" ? 4u : 2u);
        if (rawbuf.size() % es != 0u) { FAIL("input size is not a multiple of the sample size"); }
        i1.f64 = raw_type == "f64"; i1.fortran = false; i1.shape.assign(1, rawbuf.size() / es); i1.n = rawbuf.size() / es;
        if (raw_type == "s16")                                                               // the 16-bit PCM path, no header: -r gives the rate
        {
            is_wav = true; wav.rate = (unsigned)sr; wav.channels = 1;
            wav.x.resize(i1.n);
            if (i1.n) { memcpy(wav.x.data(), rawbuf.data(), rawbuf.size()); }
        }
    }
    else if (is_wav)
    {
        string why;
        if (!read_wav(is, magic, (unsigned)channel, wav, why)) { FAIL("problem reading .wav input: " + why); }
        if (have_r && sr != (double)wav.rate) { FAIL("srate (sample rate) does not match the .wav header (" + to_string(wav.rate) + " Hz)"); }
        sr = (double)wav.rate;
        i1.f64 = false; i1.fortran = false; i1.shape.assign(1, wav.x.size()); i1.n = wav.x.size();
    }
    else if (!read_npy_header(is, i1, magic)) { FAIL("problem reading header for input file"); }
    if (TOOL == TOOL_FBANK && i1.f64) { FAIL("input data type must be in {1}"); }            // srci/kaldi_fbank.cpp:6

    // options: src/kaldi_mfcc.cpp:179-246
    if (sr < DBL_EPSILON) { FAIL("srate (sample rate) must be nonnegative"); }
    if (fl < DBL_EPSILON) { FAIL("frame length must be positive"); }
    if (shft < DBL_EPSILON) { FAIL("frame shift must be positive"); }
    if (d < 0.0) { FAIL("d must be nonnegative"); }
    if (p < 0.0 || p > 1.0) { FAIL("p must be in [0.0 1.0]"); }
    for (auto &ch : wintype) { ch = (char)tolower((unsigned char)ch); }
    size_t B = 23, C = 13;
    if (TOOL != TOOL_SPECTROGRAM)
    {
        if (lof < 0.0) { FAIL("lof must be nonnegative"); }
        const double hif_arg = hif;
        if (hif < DBL_EPSILON) { hif = 0.5 * sr + hif; }                                     // offset from Nyquist
        if (hif <= lof && !(list_mode && !have_r)) { FAIL("hif must be > lof"); }            // (a list without -r: checked once the first header gives the rate)
        if (list_mode) { hif = hif_arg; }
        if (Bi < 1) { FAIL("B must be positive"); }
        B = (size_t)Bi;
    }
    if (TOOL == TOOL_MFCC || TOOL == TOOL_PLP)
    {
        if (Ci < 1) { FAIL("C must be positive"); }
        C = (size_t)Ci;
        if (C > B) { FAIL("C (num ceps) must be <= B (num fbank bins)"); }
        if (Q < 0.0) { FAIL("Q must be nonnegative"); }
    }

    // dither: the reference seeds its generator from the wall clock on every run (c/kaldi_fbank.c:97-101), and so does this tool
    // unless --seed pins the noise; --utt pins the utterance number as well (speech_hip.h, speech_set_dropin_dither)
    if (!have_seed)
    {
        timespec now; clock_gettime(CLOCK_REALTIME, &now);
        seed = (unsigned long long)now.tv_sec * 1000000000ull + (unsigned long long)now.tv_nsec;
    }
    speech_set_dropin_dither(seed, have_seed ? (long long)utt : -1);
    auto make_opts = [&](speech_opts &o) {
        speech_opts_default(&o, TOOL == TOOL_SPECTROGRAM ? SPEECH_SPECTROGRAM : TOOL == TOOL_FBANK ? SPEECH_FBANK : TOOL == TOOL_MFCC ? SPEECH_MFCC : SPEECH_PLP);
        o.dtype = SPEECH_F32; o.sr = sr; o.frame_length = fl; o.frame_shift = shft; o.snip_edges = snipe; o.dither = d; o.dc0 = dc0; o.raw_energy = rawe;
        o.preemph = p; o.mn0 = mn0; o.amp = amp; o.lof = lof; o.hif = hif; o.B = B; o.lg = lg; o.use_energy = usee; o.C = C; o.Q = Q;
        o.compress = compress; o.cep_scale = cep_scale; o.lpc_order = 0; o.dither_seed = seed;
        strncpy(o.win_type, wintype.c_str(), sizeof(o.win_type) - 1); o.win_type[sizeof(o.win_type) - 1] = '\0';
    };
    if (list_mode)
    {
        if (!ark_key.empty()) { FAIL("--wav_scp takes the utterance ids from the list, not from --ark"); }
        if (!have_o || ofile.empty() || ofile == "-") { FAIL("--wav_scp needs an output archive (-o)"); }
        ofstream ofs(ofile, append ? (ios::binary | ios::app) : (ios::binary | ios::trunc));
        if (!ofs) { FAIL("problem opening output file 1"); }
        long long off = 0;
        if (append) { ofs.seekp(0, ios::end); off = (long long)ofs.tellp(); }
        speech_opts o; make_opts(o);
        timespec tic, toc;
        clock_gettime(CLOCK_REALTIME, &tic);
        if (run_wav_scp(wav_scp, o, have_r, (unsigned)channel, ofs, off, ofile, scp_file, ((size_t)64 << 20) * (devices.empty() ? 1u : devices.size()), devices)) { return 1; }
        clock_gettime(CLOCK_REALTIME, &toc);
        if (TOOL != TOOL_SPECTROGRAM) { cerr << "elapsed time = " << double(toc.tv_sec - tic.tv_sec) * 1e3 + double(toc.tv_nsec - tic.tv_nsec) / 1e6 << " ms" << endl; }
        return 0;
    }

    // checks: src/kaldi_mfcc.cpp:250-251
    size_t big = 0;
    for (size_t s : i1.shape) { if (s > 1) { ++big; } }
    if (big > 1) { FAIL("input (X) must be a vector"); }
    const size_t N = i1.shape.empty() ? 1 : i1.n;
    if (N == 0) { FAIL("input (X) found to be empty"); }

    const bool ark = !ark_key.empty();
    if (ark && (ark_key.find_first_of(" \t\n") != string::npos)) { FAIL("ark utterance id must not contain white space"); }
    if (!ark && (append || !scp_file.empty())) { FAIL("--append and --scp need --ark=<utt-id>"); }
    if (!scp_file.empty() && (stdo1 || !wo1)) { FAIL("--scp needs an output file (-o), not stdout"); }

    // output shape: src/kaldi_mfcc.cpp:255-261.  L, stp, nfft, D and W come from the library (computed there in the working
    // precision of the input, as the reference does), so the header written below always matches what the call fills in.
    size_t L = 0, stp = 0, nfft = 0, D = 0, W = 0;
    {
        speech_opts o; make_opts(o);
        o.dtype = i1.f64 ? SPEECH_F64 : SPEECH_F32;
        speech_plan *probe = nullptr;
        if (speech_plan_create(&probe, &o, 0)) { FAIL("problem during function call"); }
        speech_plan_sizes(probe, &L, &stp, &nfft, &D, nullptr);
        const int bad = speech_plan_num_frames(probe, N, &W);
        speech_plan_destroy(probe);
        if (bad) { FAIL("frame length must be < signal length if snip_edges, and frame shift at least one sample"); }
    }

    ofstream ofs1;
    long long ark_off = 0;
    if (wo1 && !stdo1)
    {
        ofs1.open(ofile, (ark && append) ? (ios::binary | ios::app) : (ios::binary | ios::trunc));
        if (!ofs1) { FAIL("problem opening output file 1"); }
        if (ark && append) { ofs1.seekp(0, ios::end); ark_off = (long long)ofs1.tellp(); }
    }
    ostream &os = stdo1 ? cout : ofs1;
    if (wo1 && ark)
    {
        put_ark_header(os, ark_key, i1.f64, W, D);
        ark_off += (long long)ark_key.size() + 1;                                            // the scp offset points at the "\0B"
        if (!os) { FAIL("problem writing header for output file 1"); }
    }
    else if (wo1 && !raw_mode && !write_npy_header(os, i1.f64, i1.fortran, i1.fortran ? D : W, i1.fortran ? W : D)) { FAIL("problem writing header for output file 1"); }

    timespec tic, toc;
    clock_gettime(CLOCK_REALTIME, &tic);
    int rc;
    if (is_wav)
    {
        speech_opts o; make_opts(o);
        vector<float> Y(W * D + 1);
        rc = speech_extract_pcm16(&o, wav.x.data(), N, Y.data());
        if (rc) { FAIL("problem during function call"); }
        if (wo1 && !os.write((const char *)Y.data(), (streamsize)(W * D * sizeof(float)))) { FAIL("problem writing output file (Y)"); }
    }
    else if (!i1.f64)
    {
        vector<float> X(N), Y(W * D + 1);
        if (raw_mode) { memcpy(X.data(), rawbuf.data(), N * sizeof(float)); }
        else if (!is.read((char *)X.data(), (streamsize)(N * sizeof(float)))) { FAIL("problem reading input file (X)"); }
        if (TOOL == TOOL_SPECTROGRAM) { rc = codee::kaldi_spectrogram_s(Y.data(), X.data(), N, (float)sr, (float)fl, (float)shft, snipe, (float)d, dc0, rawe, (float)p, wintype.c_str(), mn0); }
        else if (TOOL == TOOL_FBANK) { rc = codee::kaldi_fbank_s(Y.data(), X.data(), N, (float)sr, (float)fl, (float)shft, snipe, (float)d, dc0, rawe, (float)p, wintype.c_str(), amp, (float)lof, (float)hif, B, lg, usee, mn0); }
        else if (TOOL == TOOL_MFCC) { rc = codee::kaldi_mfcc_s(Y.data(), X.data(), N, (float)sr, (float)fl, (float)shft, snipe, (float)d, dc0, rawe, (float)p, wintype.c_str(), (float)lof, (float)hif, B, C, (float)Q, usee, mn0); }
        else { rc = codee::kaldi_plp_s(Y.data(), X.data(), N, (float)sr, (float)fl, (float)shft, snipe, (float)d, dc0, rawe, (float)p, wintype.c_str(), (float)lof, (float)hif, B, (float)compress, C, (float)Q, (float)cep_scale, usee, mn0); }
        if (rc) { FAIL("problem during function call"); }
        if (wo1 && !os.write((const char *)Y.data(), (streamsize)(W * D * sizeof(float)))) { FAIL("problem writing output file (Y)"); }
    }
    else
    {
        vector<double> X(N), Y(W * D + 1);
        if (raw_mode) { memcpy(X.data(), rawbuf.data(), N * sizeof(double)); }
        else if (!is.read((char *)X.data(), (streamsize)(N * sizeof(double)))) { FAIL("problem reading input file (X)"); }
        if (TOOL == TOOL_SPECTROGRAM) { rc = codee::kaldi_spectrogram_d(Y.data(), X.data(), N, sr, fl, shft, snipe, d, dc0, rawe, p, wintype.c_str(), mn0); }
        else if (TOOL == TOOL_FBANK) { rc = codee::kaldi_fbank_d(Y.data(), X.data(), N, sr, fl, shft, snipe, d, dc0, rawe, p, wintype.c_str(), amp, lof, hif, B, lg, usee, mn0); }
        else if (TOOL == TOOL_MFCC) { rc = codee::kaldi_mfcc_d(Y.data(), X.data(), N, sr, fl, shft, snipe, d, dc0, rawe, p, wintype.c_str(), lof, hif, B, C, Q, usee, mn0); }
        else { rc = codee::kaldi_plp_d(Y.data(), X.data(), N, sr, fl, shft, snipe, d, dc0, rawe, p, wintype.c_str(), lof, hif, B, compress, C, Q, cep_scale, usee, mn0); }
        if (rc) { FAIL("problem during function call"); }
        if (wo1 && !os.write((const char *)Y.data(), (streamsize)(W * D * sizeof(double)))) { FAIL("problem writing output file (Y)"); }
    }
    os.flush();
    if (!scp_file.empty())
    {
        ofstream scp(scp_file, ios::app);
        scp << ark_key << ' ' << ofile << ':' << ark_off << '\n';
        if (!scp) { FAIL("problem writing scp file"); }
    }
    clock_gettime(CLOCK_REALTIME, &toc);
    if (TOOL != TOOL_SPECTROGRAM)                                                            // generated with -t: src/kaldi_mfcc.cpp:321-322; the spectrogram tool is not (Makefile:47)
    {
        cerr << "elapsed time = " << double(toc.tv_sec - tic.tv_sec) * 1e3 + double(toc.tv_nsec - tic.tv_nsec) / 1e6 << " ms" << endl;
    }
    return 0;
}
