/* zlhip_voice_adapter.h -- the JUCE SynthesiserVoice-shaped surface of libzl's sampler voice on top of the engine.
 *
 * In the reference, `class SamplerSynthVoice : public QObject, public juce::SamplerVoice`
 * (lib/SamplerSynthVoice.h:9-35) carries one voice's state and renders it in process(); SamplerChannel owns eight of
 * them, adds them to a juce::Synthesiser and starts them through juce::Synthesiser::startVoice
 * (lib/SamplerSynth.cpp:178-185,204-215).  With the MI355X engine the state and the rendering of every voice live on
 * the device; what remains of the class is its callback surface.  This header keeps that surface -- same method
 * names, argument orders and meaning -- so that code written against SamplerSynthVoice (the JUCE Synthesiser, the
 * channel's command handling) keeps compiling, and forwards each call to the C-ABI of include/zlhip.h:
 *
 *   canPlaySound(sound)                 -> dynamic type check, as :53-56
 *   setCurrentCommand(cmd)              -> idle voice: remembered until startNote (:94-96)
 *                                          playing voice: zlhip_update_voice (merge of :59-93)
 *   setStartTick(tick)                  -> remembered until startNote (:100-103)
 *   startNote(note, velocity, sound, _) -> zlhip_start_voice on this voice's (bus, slot) (:110-144)
 *   stopNote(velocity, allowTailOff)    -> zlhip_stop_voice (:146-169)
 *   pitchWheelMoved / controllerMoved   -> no-ops, as :171-172
 *   process(...)                        -> nothing to do per voice: SamplerChannel::process calls zlhip_render once for
 *                                          all channels; kept so call sites compile
 *   renderNextBlock(buffer, start, num) -> the juce::SynthesiserVoice rendering callback (juce::SamplerVoice overrides it;
 *                                          the reference's class inherits that override, lib/SamplerSynthVoice.h:9-29, and
 *                                          never reaches it: nothing calls Synthesiser::renderNextBlock).  A host that DOES
 *                                          drive the JUCE Synthesiser gets the engine's result through it: the channel
 *                                          publishes the block zlhip_render produced for this bus (BusBlock), voice slot 0
 *                                          adds it into the output buffer, the other slots add nothing -- the Synthesiser
 *                                          calls every voice, so the buffer receives the bus mix exactly once
 *   isPlaying                           -> refreshed from the engine by syncFromEngine()
 *
 * The class is a template over the JUCE / libzl types it touches, so this header needs no JUCE include itself and is
 * compile-checked in this repository against minimal test doubles (tests/cpp/adapter_check.cpp).  Inside libzl:
 *
 *   struct ZlCommandFields {   // how libzl's ClipCommand (lib/ClipCommand.h:11-32) maps onto zlhip_clip_command
 *       static void fill(const ClipCommand &c, int engineClipId, zlhip_clip_command &o);
 *   };
 *   using SamplerSynthVoice = zlhip::VoiceAdapter<juce::SamplerVoice, juce::SynthesiserSound, SamplerSynthSound,
 *                                                 ClipCommand, ZlCommandFields, juce::AudioBuffer<float>>;
 */
#ifndef ZLHIP_VOICE_ADAPTER_H
#define ZLHIP_VOICE_ADAPTER_H

#include <cstdint>
#include "zlhip.h"

namespace zlhip {

/* The block zlhip_render produced for one bus, published by the channel once per cycle (host memory, [nframes] each). */
struct BusBlock {
    const float *left = nullptr, *right = nullptr;
    int nframes = 0;
};

/* SoundT must offer `int engineClipId() const` (the id zlhip_sound_upload returned for the clip's source).
 * AudioBufferT is juce::AudioBuffer<float> (used: getNumChannels(), addFrom(channel, destStart, const float *, num)). */
template <class SamplerVoiceBase, class SynthesiserSoundT, class SoundT, class ClipCommandT, class CommandFields, class AudioBufferT>
class VoiceAdapter : public SamplerVoiceBase {
public:
    VoiceAdapter(zlhip_engine *engine, int bus, int slot, const BusBlock *busBlock = nullptr)
        : engine_(engine), bus_(bus), slot_(slot), busBlock_(busBlock) {}
    ~VoiceAdapter() override {}

    bool canPlaySound(SynthesiserSoundT *sound) override { return dynamic_cast<const SoundT *>(sound) != nullptr; }

    void setCurrentCommand(ClipCommandT *clipCommand)
    {
        if (!clipCommand) return;
        if (isPlaying && sound_) {
            zlhip_clip_command c;
            zlhip_clip_command_clear(&c);
            CommandFields::fill(*clipCommand, sound_->engineClipId(), c);
            (void)zlhip_update_voice(engine_, bus_, slot_, &c);
        } else {
            command_ = clipCommand;
        }
    }
    ClipCommandT *currentCommand() const { return command_; }

    void setStartTick(uint64_t startTick) { startTick_ = startTick; }

    void startNote(int midiNoteNumber, float velocity, SynthesiserSoundT *s, int /*currentPitchWheelPosition*/) override
    {
        SoundT *sound = dynamic_cast<SoundT *>(s);
        if (!sound || !command_) return;
        zlhip_clip_command c;
        zlhip_clip_command_clear(&c);
        CommandFields::fill(*command_, sound->engineClipId(), c);
        c.midi_note = midiNoteNumber;
        c.volume = velocity;                                      /* lgain = rgain = velocity, :131-132 */
        c.start_playback = 1;
        if (zlhip_start_voice(engine_, bus_, slot_, &c, startTick_) == 1) { isPlaying = true; sound_ = sound; }
    }

    void stopNote(float /*velocity*/, bool allowTailOff) override
    {
        (void)zlhip_stop_voice(engine_, bus_, slot_, allowTailOff ? 1 : 0);
        if (!allowTailOff) { isPlaying = false; command_ = nullptr; sound_ = nullptr; }
    }

    void pitchWheelMoved(int /*newValue*/) override {}
    void controllerMoved(int /*controllerNumber*/, int /*newValue*/) override {}

    /* per-voice rendering happens on the device, for every voice of every channel at once */
    void process(float * /*leftBuffer*/, float * /*rightBuffer*/, uint32_t /*nframes*/, uint32_t /*current_frames*/,
                 uint64_t /*current_usecs*/, uint64_t /*next_usecs*/, float /*period_usecs*/) {}

    /* juce::SynthesiserVoice::renderNextBlock: ADDS this voice's output to outputBuffer[startSample, startSample + numSamples).
     * The voices of a bus are mixed on the device, so slot 0 delivers the whole bus and the other slots deliver silence. */
    void renderNextBlock(AudioBufferT &outputBuffer, int startSample, int numSamples) override
    {
        if (slot_ != 0 || !busBlock_ || !busBlock_->left || startSample < 0 || numSamples <= 0) return;
        if (startSample + numSamples > busBlock_->nframes) return;          /* the channel has not rendered this range */
        const int channels = outputBuffer.getNumChannels();
        if (channels > 0) outputBuffer.addFrom(0, startSample, busBlock_->left + startSample, numSamples);
        if (channels > 1) outputBuffer.addFrom(1, startSample, busBlock_->right + startSample, numSamples);
    }

    /* a voice that ended on the device (tail finished, one-shot reached its stop position) frees itself here, as
     * stopNote(0, false) does inside the reference's process() (:249-252,258-261); call once per rendered block */
    void syncFromEngine()
    {
        if (isPlaying && zlhip_voice_is_playing(engine_, bus_, slot_) == 0) { isPlaying = false; command_ = nullptr; sound_ = nullptr; }
    }

    bool isPlaying = false;

private:
    zlhip_engine *engine_;
    int bus_, slot_;
    const BusBlock *busBlock_;
    ClipCommandT *command_ = nullptr;
    SoundT *sound_ = nullptr;
    uint64_t startTick_ = 0;
};

}  /* namespace zlhip */
#endif
