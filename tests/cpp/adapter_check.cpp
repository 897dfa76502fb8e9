// Compile- and run-check of include/zlhip_voice_adapter.h against minimal test doubles of the JUCE / libzl types it
// is written for (the shapes of juce::SynthesiserVoice's virtuals, SamplerSynthSound, ClipCommand).  Test
// infrastructure: the doubles only give the template something to derive from; nothing of JUCE is restated.
//
// Scenario: two voices are driven through the adapter's SynthesiserVoice surface (setCurrentCommand, setStartTick,
// startNote, a live volume update, stopNote with and without tail-off); a second engine gets the same musical events
// through the channel-level command API (zlhip_handle_command).  Both must render the same bits.
// exit codes: 0 ok, 1 mismatch / failure, 77 no HIP device (the CPU tier only checks that this file builds and links)
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "zlhip_voice_adapter.h"

struct FakeSynthesiserSound { virtual ~FakeSynthesiserSound() {} };
struct FakeAudioBuffer {                                           // the two juce::AudioBuffer<float> members the adapter uses
    std::vector<std::vector<float>> ch;
    FakeAudioBuffer(int channels, int n) : ch((size_t)channels, std::vector<float>((size_t)n, 0.0f)) {}
    int getNumChannels() const { return (int)ch.size(); }
    void addFrom(int destChannel, int destStartSample, const float *source, int numSamples)
    {
        for (int i = 0; i < numSamples; ++i) ch[(size_t)destChannel][(size_t)(destStartSample + i)] += source[i];
    }
};
struct FakeSamplerVoice {                                          // the virtuals SamplerSynthVoice overrides / inherits
    virtual ~FakeSamplerVoice() {}
    virtual void renderNextBlock(FakeAudioBuffer &, int startSample, int numSamples) = 0;
    virtual bool canPlaySound(FakeSynthesiserSound *) = 0;
    virtual void startNote(int midiNoteNumber, float velocity, FakeSynthesiserSound *, int currentPitchWheelPosition) = 0;
    virtual void stopNote(float velocity, bool allowTailOff) = 0;
    virtual void pitchWheelMoved(int) = 0;
    virtual void controllerMoved(int, int) = 0;
};
struct FakeSound : FakeSynthesiserSound { int id = -1; int engineClipId() const { return id; } };
struct OtherSound : FakeSynthesiserSound {};
struct FakeClipCommand { int midiNote = 60, midiChannel = -2; bool looping = false, changeVolume = false; float volume = 1.0f; };
struct Fields {
    static void fill(const FakeClipCommand &c, int clip, zlhip_clip_command &o)
    {
        o.clip = clip; o.midi_note = c.midiNote; o.midi_channel = c.midiChannel; o.looping = c.looping ? 1 : 0;
        o.change_looping = 1; o.change_volume = c.changeVolume ? 1 : 0; o.volume = c.volume;
    }
};
using Voice = zlhip::VoiceAdapter<FakeSamplerVoice, FakeSynthesiserSound, FakeSound, FakeClipCommand, Fields, FakeAudioBuffer>;

static zlhip_engine *make_engine(const std::vector<float> &L, const std::vector<float> &R, int *clip)
{
    zlhip_config cfg; zlhip_config_default(&cfg);
    cfg.num_buses = 2; cfg.voices_per_bus = 4; cfg.max_frames = 128; cfg.max_batch_blocks = 1; cfg.max_sounds = 4;
    cfg.sound_arena_bytes = 1 << 20;
    zlhip_engine *e = nullptr;
    int rc = zlhip_engine_create(&cfg, &e);
    if (rc == ZLHIP_ERR_NO_DEVICE) { std::printf("no HIP device: build and link check only\n"); std::exit(77); }
    if (rc != ZLHIP_OK) { std::printf("engine_create failed: %s\n", zlhip_strerror(rc)); std::exit(1); }
    if (zlhip_sound_upload(e, L.data(), R.data(), (int32_t)L.size(), 48000.0, clip) != ZLHIP_OK) std::exit(1);
    zlhip_clip_params p; zlhip_clip_params_default(&p, (float)L.size() / 48000.0f);
    p.length_in_beats = 0.75f; p.length_seconds = 0.02f; p.pan = 0.3f; p.adsr_release = 0.004f;
    if (zlhip_clip_set(e, *clip, &p) != ZLHIP_OK) std::exit(1);
    return e;
}

static float g_l[2 * 128], g_r[2 * 128];                          // the last block of the adapter-driven engine, [bus][frame]
static void render(zlhip_engine *e, int k, std::vector<float> &out, bool keep = false)
{
    zlhip_clock ck; std::memset(&ck, 0, sizeof ck);
    ck.current_usecs = (uint64_t)k * 2667; ck.next_usecs = (uint64_t)(k + 1) * 2667; ck.jack_subbeat_length_usecs = 5208;
    float l[2 * 128], r[2 * 128];
    if (zlhip_render(e, 128, &ck, l, r) != ZLHIP_OK) { std::printf("render failed: %s\n", zlhip_last_error(e)); std::exit(1); }
    out.insert(out.end(), l, l + 256); out.insert(out.end(), r, r + 256);
    if (keep) { std::memcpy(g_l, l, sizeof l); std::memcpy(g_r, r, sizeof r); }
}

int main()
{
    std::vector<float> L(3000), R(3000);
    for (size_t i = 0; i < L.size(); ++i) { L[i] = std::sin(0.05f * (float)i); R[i] = std::cos(0.031f * (float)i); }
    int clipA = -1, clipB = -1;
    zlhip_engine *ea = make_engine(L, R, &clipA);                 // driven through the voice adapter
    zlhip_engine *eb = make_engine(L, R, &clipB);                 // driven through channel commands
    FakeSound sound; sound.id = clipA;
    OtherSound other;
    zlhip::BusBlock block0;                                        // what the channel of bus 0 publishes per cycle
    Voice v0(ea, 0, 0, &block0), v1(ea, 0, 1, &block0);
    if (!v0.canPlaySound(&sound) || v0.canPlaySound(&other)) { std::printf("canPlaySound wrong\n"); return 1; }

    FakeClipCommand c0; c0.midiNote = 60; c0.looping = true;
    FakeClipCommand c1; c1.midiNote = 67; c1.looping = true;
    std::vector<float> a, b;
    // block 0: both voices start (velocity = the command's volume in the reference, SamplerSynth.cpp:210)
    v0.setCurrentCommand(&c0); v0.setStartTick(0); v0.startNote(60, 0.7f, &sound, 0);
    v1.setCurrentCommand(&c1); v1.setStartTick(0); v1.startNote(67, 0.4f, &sound, 0);
    zlhip_clip_command k; zlhip_clip_command_clear(&k);
    k.clip = clipB; k.midi_channel = -2; k.start_playback = 1; k.looping = 1; k.change_looping = 1;
    k.midi_note = 60; k.volume = 0.7f; if (zlhip_handle_command(eb, &k, 0) != 1) return 1;
    k.midi_note = 67; k.volume = 0.4f; if (zlhip_handle_command(eb, &k, 0) != 1) return 1;
    for (int i = 0; i < 3; ++i) { render(ea, i, a, true); render(eb, i, b); }
    {
        // the juce::Synthesiser rendering callback: Synthesiser::renderVoices calls EVERY voice with the same buffer and
        // range; the buffer must receive bus 0's mix exactly once, only inside the range, added to what it held
        block0.left = g_l; block0.right = g_r; block0.nframes = 128;
        FakeAudioBuffer buf(2, 128);
        for (auto &c : buf.ch) for (float &x : c) x = 0.25f;
        FakeSamplerVoice *voices[2] = { &v0, &v1 };
        for (FakeSamplerVoice *v : voices) v->renderNextBlock(buf, 16, 96);
        double energy = 0.0;
        for (int i = 0; i < 128; ++i) {
            const bool in = i >= 16 && i < 112;
            const float wl = 0.25f + (in ? g_l[i] : 0.0f), wr = 0.25f + (in ? g_r[i] : 0.0f);
            if (buf.ch[0][(size_t)i] != wl || buf.ch[1][(size_t)i] != wr) { std::printf("renderNextBlock: frame %d wrong\n", i); return 1; }
            energy += (double)g_l[i] * g_l[i];
        }
        if (energy == 0.0) { std::printf("renderNextBlock: the block was silent\n"); return 1; }
        FakeAudioBuffer mono(1, 128);
        v0.renderNextBlock(mono, 0, 128); v0.renderNextBlock(mono, 100, 64);   // second call: out of the rendered range, ignored
        for (int i = 0; i < 128; ++i) if (mono.ch[0][(size_t)i] != g_l[i]) { std::printf("renderNextBlock: mono buffer wrong\n"); return 1; }
    }
    // block 3: live volume change on voice 0 (setCurrentCommand on a playing voice)
    FakeClipCommand upd; upd.midiNote = 60; upd.looping = true; upd.changeVolume = true; upd.volume = 0.25f;
    v0.setCurrentCommand(&upd);
    zlhip_clip_command_clear(&k); k.clip = clipB; k.midi_channel = -2; k.midi_note = 60; k.change_volume = 1; k.volume = 0.25f;
    k.change_looping = 1; k.looping = 1;
    if (zlhip_handle_command(eb, &k, 0) != 1) return 1;
    for (int i = 3; i < 5; ++i) { render(ea, i, a); render(eb, i, b); }
    // block 5: voice 1 stops with its release tail
    v1.stopNote(0.0f, true);
    zlhip_clip_command_clear(&k); k.clip = clipB; k.midi_channel = -2; k.midi_note = 67; k.stop_playback = 1;
    (void)zlhip_handle_command(eb, &k, 0);
    for (int i = 5; i < 9; ++i) { render(ea, i, a); render(eb, i, b); v0.syncFromEngine(); v1.syncFromEngine(); }
    if (v1.isPlaying) { std::printf("voice 1 should have freed itself after its tail\n"); return 1; }
    if (!v0.isPlaying) { std::printf("voice 0 should still play\n"); return 1; }
    if (a.size() != b.size() || std::memcmp(a.data(), b.data(), a.size() * sizeof(float)) != 0) { std::printf("adapter and command paths differ\n"); return 1; }
    // block 9: voice 0 is cut without tail-off: silence from the next block on
    v0.stopNote(0.0f, false);
    std::vector<float> tail;
    render(ea, 9, tail);
    double energy = 0.0, before = 0.0;
    for (float x : tail) energy += (double)x * x;
    for (size_t i = a.size() - 512; i < a.size(); ++i) before += (double)a[i] * a[i];
    if (energy != 0.0 || before == 0.0) { std::printf("hard stop did not silence the voice (%g, %g)\n", energy, before); return 1; }
    zlhip_engine_destroy(ea); zlhip_engine_destroy(eb);
    std::printf("voice adapter ok: %zu samples identical through both surfaces\n", a.size());
    return 0;
}
