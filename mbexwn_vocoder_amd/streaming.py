"""Chunked (streaming) mel inversion that reproduces the whole-utterance output.

The reference has no streaming mode: its graph is non-causal (``padding="SAME"``) and the author notes that causal
operation "would require a dedicated implementation" (reference custom_pulsed_generator.py:213-217).  This module
serves BASELINE config 5 (many concurrent streams, short ticks) on top of the offline engine:

* every op of the graph has a finite receptive field, so the audio of mel frames [t0, t0+chunk) only depends on
  the mel frames [t0-LEFT, t0+chunk+RIGHT): a tick runs the engine on that window and keeps the middle;
* the one unbounded dependency is the wavetable phase accumulator (float32 running sum in 1000-sample chunks,
  reference tf_wavetable.py:429-492): its state (running sum of the chunk in progress, un-wrapped offset sum,
  position in the chunk) is carried from tick to tick through ``mbx_forward_stream`` -- the phase inside a window is
  then bit-identical to the offline phase;
* at the true start / end of an utterance the window edge IS the utterance edge, so the reference's boundary
  semantics (symmetric / zero padding, un-normalised head and tail of the inverse STFT) apply where they should.

Receptive field in mel frames (canonical model): F0-net 3 convs k=3 (+-3) and the interpolator (+1); the phase needs
valid F0, so pulses are valid from window frame 4; WaveNet (dilations 1..16, k=3: +-31 steps, + 9 steps of conditioning
interpolation towards a row that a region's end clamps = 2 frames) -> 6;
PQMF (+-4 steps) -> 7; STFT frame + overlap-add (-3 / +4 frames) -> LEFT = 10, RIGHT = 11 (look-ahead 137.5 ms).
``StreamingSynthesizer`` derives the margins from the model configuration.
"""
import numpy as np


def pack_state(cum=0.0, offset_sum=0.0, pos_in_chunk=0, start_sample=0, save_sample=-1):
    """One ``mbx_stream_state`` as 6 int32 words (floats bit-cast)."""
    ff = np.asarray([cum, offset_sum], dtype=np.float32).view(np.int32)
    return np.asarray([ff[0], ff[1], pos_in_chunk, start_sample, save_sample, 0], dtype=np.int32)


def norm_reach(dims, config):
    """Reach in mel frames of the optional RMS normalisation of the mel input (reference wavegen_1d.py:697-726): every
    smoothing iteration overlap-adds the per-frame RMS with the smoothing window and re-estimates it through the analysis
    window, i.e. frame t then depends on the frames within (smooth_win + win) / (2 hop) of it; 0 without normalisation."""
    if not dims.normalize_rms_from_mell:
        return 0
    from .norm_mel import NormMel
    nm = NormMel(config)
    return nm.iters * ((nm.smooth_win_size + nm.win) // (2 * nm.hop))


def cond_chain_reach(dims):
    """(left, right) reach in mel frames of the conditioning chain -- the conditioning layer and the pre-conditioning
    convolutions in front of it, all with kernel size cond_kernel_size and zero SAME padding, which the library pads
    (k - 1) // 2 frames in front and k // 2 behind (csrc/mbx_api.hip, cond_chain): an even kernel size reaches one frame
    further to the right than to the left, per convolution."""
    n_cond = 0 if dims.wn_disable_conditioning else 1 + len(dims.wn_pre_cond_channels)
    return n_cond * ((dims.cond_kernel_size - 1) // 2), n_cond * (dims.cond_kernel_size // 2)


def stream_margins(dims, config):
    """(left, right, pulse_lead, act_left, act_right, wn_reach) in mel frames, from the layer geometry of the model."""
    mb = config["mbexwn_config"]
    nr = norm_reach(dims, config)      # the normalised mel of a window is reproducible nr frames inside its edges only

    def subnet_reach(specs):
        left = right = 0
        for spec in specs:
            if spec[0] == "L":
                continue
            ks = int(spec[0])
            left += (ks - 1) // 2 + ((ks - 1) % 2)
            right += (ks - 1) // 2
        return left, right

    f0_l, f0_r = subnet_reach(mb["pp_subnet"])
    f0_r += 1                                             # interpolation towards the next frame
    spf = dims.steps_per_frame
    wn_steps = sum(dims.wn_dilation(ll) * (dims.wn_kernel_size - 1) // 2 for ll in range(dims.wn_layers))
    # the conditioning is interpolated towards the next conditioning row, which at the end of a region is the clamped
    # last one: the last cond_lin_upsampling - 1 rows of every layer's gate are off, and that spreads backwards by the
    # reach of the layers behind
    wn_steps += dims.cond_lin_upsampling - 1
    wn_frames = -(-wn_steps // spf)
    pqmf_frames = -(-(int(mb["multi_band_config"]["taps"]) // 2) // dims.hop_size)
    # conditioning chain: the conditioning layer and the pre-conditioning convolutions in front of it (same kernel size,
    # zero SAME padding: (k - 1) // 2 frames to the left, k // 2 to the right, per convolution); + 1: the interpolation
    # towards the next conditioning row
    cond_l, cond_r = cond_chain_reach(dims)
    cond_r += 1
    vt_l, vt_r = (0, 0) if dims.no_envelope else subnet_reach(mb["ps_subnet"])     # cepstrum of a frame <- mel frames around it
    stft_l, stft_r = 3, 4                                  # frame t reaches excitation frames t-3 .. t+4
    # first window frame whose mel-rate inputs of the WaveNet -- F0 / phase and the conditioning rows -- are reproducible
    pulse_lead = nr + max(f0_l + 1, cond_l)
    left = pulse_lead + wn_frames + pqmf_frames + stft_l
    right = nr + max(f0_r, cond_r) + wn_frames + pqmf_frames + stft_r
    # the envelope filter of the frames around the emitted ones needs their cepstra
    left = max(left, nr + vt_l + stft_l)
    right = max(right, nr + vt_r + stft_r)
    smooth = 3                                             # F0 smoother of the lifter selection: +-3 frames of valid F0
    left = max(left, pulse_lead + smooth + 1)
    right = max(right, nr + f0_r + smooth + 2)
    # margins of the stages from the WaveNet on (the active region of a window, mbx_forward_options.active_begin)
    act_left = wn_frames + pqmf_frames + stft_l
    act_right = wn_frames + pqmf_frames + stft_r
    return left, right, pulse_lead, act_left, act_right, wn_frames


def frontend_reach(dims, config):
    """(left, right) reach in mel frames of the mel-rate front end (F0-net, VTF-net, conditioning convolution): the output
    of frame t depends on the mel frames [t - left, t + right]."""
    mb = config["mbexwn_config"]

    def reach(specs):
        left = right = 0
        for spec in specs:
            if spec[0] == "L" or (isinstance(spec[0], str) and spec[0].startswith("L")):
                continue
            ks = int(spec[0])
            left += (ks - 1) // 2 + ((ks - 1) % 2)
            right += (ks - 1) // 2
        return left, right

    f0_l, f0_r = reach(mb["pp_subnet"])
    vt_l, vt_r = reach(mb["ps_subnet"])
    # the conditioning layer and the pre-conditioning convolutions in front of it (same kernel size, zero SAME padding)
    ck_l, ck_r = cond_chain_reach(dims)
    # + 1: the interpolators (F0 contour, conditioning rows) reach the next frame
    return max(f0_l, vt_l, ck_l), max(f0_r, vt_r, ck_r) + 1


class _Stream:
    def __init__(self):
        # the frames a stream has received live in the synthesizer's shared input buffers (row `slot`): absolute frame f
        # sits at column f - base; frames in front of the next window are dropped when the row runs full
        self.base = 0             # absolute frame at column 0 of the stream's row
        self.have = 0             # frames received so far (absolute)
        self.emitted = 0          # frames of audio already produced
        self.closed = False
        self.slot = -1            # row of the synthesizer's sub-band store
        self.carry_pos = None     # the store holds the sub-bands of frames [carry_pos - sr_left, + carry_frames)
        self.carry_frames = 0
        self.layer_end = None     # the layer store holds the WaveNet state of a region that ended at this frame
        # phase state valid just in front of absolute pulse sample `state_frame * pulse_per_frame`
        self.state = (0.0, 0.0, 0)
        self.state_frame = 0
        self.ticks = 0            # chunks emitted so far: position in the synthesizer's tick schedule


class StreamingSynthesizer:
    """Serves any number of concurrent streams with one batched engine call per tick."""

    def __init__(self, engine, chunk_frames=8):
        """``chunk_frames``: frames a stream emits per tick -- an int, or a cyclic schedule of ints for tick lengths that
        are not a whole number of frames: BASELINE config 5's 80 ms are 6.4 frames of 12.5 ms, which the schedule
        (6, 6, 7, 6, 7) delivers exactly on average (32 frames = 400 ms per period).  Every stream walks the schedule from
        its own first tick.  The geometry of a steady tick (position of the aligned window start relative to the emitted
        frames, window length) repeats with the period of the schedule when that period is a multiple of the window
        alignment (8 frames for the canonical model: 8-frame ticks, or the 32-frame period of the 80 ms schedule): each
        phase of the period then has its own captured hipGraph, all of them working on one device-resident window; any
        other schedule runs launch by launch (still with the carried sub-bands, per-layer state and phase)."""
        self.engine = engine
        self.dims = engine.dims
        # stream windows run the float32 F(2,3) / direct gate kernels whatever the handle's precision: an engine with the
        # opt-in split half precision would synthesise offline with other kernels than its streams, and the documented
        # bit-equality between a stream and the offline synthesis would silently not hold
        info = engine.conv_form_info()
        if info.get("split_f16_layers", 0) > 0 or info.get("split_f16_gate_layers", 0) > 0:
            raise ValueError("StreamingSynthesizer needs a float32 engine: this one runs its whole-item forwards in split half "
                             "precision (precision='split_f16'), streams would not be bit-equal to its offline synthesis")
        self.schedule = [int(chunk_frames)] if np.isscalar(chunk_frames) else [int(cc) for cc in chunk_frames]
        if not self.schedule or min(self.schedule) < 1:
            raise ValueError("chunk_frames must be a positive int or a non-empty schedule of positive ints")
        self.uniform = len(set(self.schedule)) == 1
        if self.uniform:
            self.schedule = self.schedule[:1]
        self.chunk = self.schedule[0] if self.uniform else min(self.schedule)
        (self.left, self.right, self.lead, self.act_left, self.act_right,
         self.wn_reach) = stream_margins(engine.dims, engine.config)
        # sub-band rows carried from tick to tick: the stages behind the WaveNet reach sr_left frames in front of the
        # emitted ones and sr_right behind them; these frames were computed exactly by the previous tick, so the WaveNet
        # of a tick only runs on the frames behind them (plus its own reach)
        self.sr_left = self.act_left - self.wn_reach
        self.sr_right = self.act_right - self.wn_reach
        self.carry = True
        self._store = None            # (slots, (sr_left + sr_right) * steps_per_frame, subbands) on the device
        # per-layer WaveNet state carried from tick to tick (mbx_forward_options.layer_store): layer l is exact up to its
        # own reach in front of layer l-1, so a steady tick runs every layer on the new rows only instead of on the
        # region [emitted, emitted + chunk + act_right) with the WaveNet's reach recomputed on both sides
        ff, reach, min_rows = engine.layer_state_info()
        self.layer_carry = ff > 0 and reach == self.wn_reach * engine.dims.steps_per_frame
        self._layer_floats, self._layer_min_rows = ff, min_rows
        self._layer_store = None      # (slots, floats per slot) on the device
        # mel-rate front end (conditioning rows, cepstrum, F0 contour) carried from tick to tick in a ring per stream: a
        # replayed steady tick runs the sub-nets only on the frames its new mel frames can reach
        self.fe_left, self.fe_right = frontend_reach(engine.dims, engine.config)
        self.fe_carry = bool(getattr(engine, "frontend_carry_supported", False)) and self.chunk >= self.fe_left
        ring = 1
        while ring < self.left + self.chunk + self.right + 2 * 16:
            ring *= 2
        self._fe_ring = ring
        self._fe_store = None         # (slots, ring frames, floats per frame) on the device
        self._free_slots = []
        # input frames of every stream, one row per slot (shared so that a tick gathers its frames with one indexed copy)
        self._in_cap = 256            # frames per row (grows on demand)
        self._in_mel = np.zeros((0, self._in_cap, self.dims.mel_channels), dtype=np.float32)
        self._in_noise = np.zeros((0, self._in_cap, self.dims.steps_per_frame), dtype=np.float32)
        # The Winograd form of the dilated convolution pairs outputs t and t+d inside blocks of 2d steps counted from
        # the first row of the item; a window that starts on a multiple of 2*d_max steps pairs exactly like the offline
        # run, which keeps the streamed audio bit-identical (any other start is equal up to float32 rounding only).
        import math
        d_max = max(engine.dims.wn_dilation(ll) for ll in range(engine.dims.wn_layers))
        self.align = (2 * d_max) // math.gcd(2 * d_max, engine.dims.steps_per_frame)
        self.streams = {}
        # ticks of a schedule whose period is a whole number of alignment steps have a geometry that repeats per phase
        self.periodic = sum(self.schedule) % self.align == 0
        # frames of the device windows of such a schedule: every phase's window fits (uniform: the window length itself)
        self._tcap = 0 if self.uniform else self.left + max(self.schedule) + self.right + self.align
        # Steady ticks (every stream continues with the geometry of the tick before) are replayed as a captured hipGraph:
        # the windows stay on the device (mbx_window_advance appends the new frames), every integer argument of the call is
        # a constant of the capture, and one graph launch stands for the ~30 kernel launches of a tick.  This is the
        # practical form of BASELINE config 5's "persistent-kernel path": the launch sequence persists, not a kernel.
        self.use_graph = True
        self._inputs_changed = True
        self._steady = None               # the steady run in progress: its streams, the recorded phases and their graphs
        self.graph_ticks = 0              # ticks served by a graph replay
        self.last_tick_replayed = False
        self.time_device = False          # bench: bracket the engine call of a tick with events on its stream
        self.last_tick_device_ms = None
        self.last_tick_frames = 0         # window frames of the last tick (all streams): mel-rate stages
        self.last_tick_active_frames = 0  # frames of the active regions: PQMF, STFT filter, overlap-add
        self.last_tick_wavenet_frames = 0 # frames the WaveNet ran on
        self.last_tick_layer_rows = 0     # > 0: steady tick (every WaveNet layer ran on that many new rows per stream only)

    @property
    def lookahead_ms(self):
        return 1000.0 * self.right * self.dims.hop_size / self.dims.sample_rate

    def _grow_inputs(self, slots, cap):
        """Make the shared input buffers at least (slots, cap frames) large, keeping their contents."""
        old_mel, old_noise = self._in_mel, self._in_noise
        if slots <= old_mel.shape[0] and cap <= self._in_cap:
            return
        slots, cap = max(slots, old_mel.shape[0]), max(cap, self._in_cap)
        self._in_mel = np.zeros((slots, cap, self.dims.mel_channels), dtype=np.float32)
        self._in_noise = np.zeros((slots, cap, self.dims.steps_per_frame), dtype=np.float32)
        self._in_mel[:old_mel.shape[0], :self._in_cap] = old_mel
        self._in_noise[:old_noise.shape[0], :self._in_cap] = old_noise
        self._in_cap = cap

    def open(self, stream_id):
        self._leave_steady()
        st = _Stream()
        import torch
        rows = (self.sr_left + self.sr_right) * self.dims.steps_per_frame
        if not self._free_slots:
            old = self._store
            n_old = 0 if old is None else int(old.shape[0])
            n_new = max(16, 2 * n_old)
            self._store = torch.zeros((n_new, rows, self.dims.subbands), dtype=torch.float32, device=self.engine.device)
            if old is not None:
                self._store[:n_old] = old
            if self.layer_carry:
                old_l = self._layer_store
                self._layer_store = torch.zeros((n_new, self._layer_floats), dtype=torch.float32, device=self.engine.device)
                if old_l is not None:
                    self._layer_store[:n_old] = old_l
            if self.fe_carry:
                old_f = self._fe_store
                self._fe_store = torch.zeros((n_new, self._fe_ring, self.engine.frontend_frame_floats), dtype=torch.float32,
                                             device=self.engine.device)
                if old_f is not None:
                    self._fe_store[:n_old] = old_f
            self._free_slots = list(range(n_new - 1, n_old - 1, -1))
            # (the stores moved: captured ticks point at the old ones -- open() has left the steady run above)
            self._grow_inputs(n_new, self._in_cap)
        st.slot = self._free_slots.pop()
        self.streams[stream_id] = st

    def close(self, stream_id):
        """Forget a finished stream (its slot of the sub-band store is reused)."""
        self._leave_steady()
        st = self.streams.pop(stream_id)
        self._free_slots.append(st.slot)

    def push(self, stream_id, mel_frames, noise=None, last=False):
        """Append mel frames (n, mel_channels) and the matching N(0,1) draw (n*steps_per_frame,) to a stream."""
        st = self.streams[stream_id]
        mel_frames = np.asarray(mel_frames, dtype=np.float32).reshape(-1, self.dims.mel_channels)
        n = mel_frames.shape[0]
        spf = self.dims.steps_per_frame
        if self.dims.noise_sigma:
            if noise is None:
                raise ValueError("noise is required (explicit input of the path)")
            noise = np.asarray(noise, dtype=np.float32).reshape(-1, spf)
            if noise.shape[0] != n:
                raise ValueError("noise must hold steps_per_frame values per pushed mel frame")
        if st.have - st.base + n > self._in_cap:
            # drop the frames no later window can reach (windows start at aligned(emitted - left); a stale `emitted` of a
            # stream inside a run of replayed ticks only keeps more than necessary), then grow the rows if that is not enough
            keep_from = max(st.base, ((st.emitted - self.left) // self.align) * self.align)
            drop, live = keep_from - st.base, st.have - keep_from
            if drop > 0:
                self._in_mel[st.slot, :live] = self._in_mel[st.slot, drop:drop + live]
                self._in_noise[st.slot, :live] = self._in_noise[st.slot, drop:drop + live]
                st.base = keep_from
            if st.have - st.base + n > self._in_cap:
                self._grow_inputs(self._in_mel.shape[0], max(2 * self._in_cap, st.have - st.base + n))
        col = st.have - st.base
        self._in_mel[st.slot, col:col + n] = mel_frames
        if self.dims.noise_sigma:
            self._in_noise[st.slot, col:col + n] = noise
        st.have += n
        st.closed = st.closed or last
        self._inputs_changed = True       # cached per-stream vectors of a run of replayed ticks are stale

    def _ready(self, st):
        have = st.have
        if st.emitted >= have:
            return 0
        chunk = self.schedule[st.ticks % len(self.schedule)]
        if st.closed:
            return min(chunk, have - st.emitted)
        return chunk if have >= st.emitted + chunk + self.right else 0

    def tick(self):
        """One batched engine call over every stream that can emit. Returns {stream_id: audio ndarray}."""
        import torch
        self.last_tick_replayed = False
        if self._steady is not None:
            status = self._steady_status()
            if status == "replay" and self.use_graph:
                return self._graph_tick()
            if status == "broken":
                self._leave_steady()
            else:                                     # a phase of the schedule that has no recorded tick yet
                self._sync_streams()
        todo = [(sid, st, self._ready(st)) for sid, st in self.streams.items()]
        todo = [(sid, st, nn) for sid, st, nn in todo if nn > 0]
        if not todo:
            return {}
        ppf, spf, hop = self.dims.pulse_per_frame, self.dims.steps_per_frame, self.dims.hop_size
        windows = []
        for sid, st, nn in todo:
            have = st.have
            ws = max(0, ((st.emitted - self.left) // self.align) * self.align)
            we = have if st.closed and st.emitted + nn + self.right >= have else st.emitted + nn + self.right
            we = min(we, have)
            windows.append((ws, we))
        tmax = max(we - ws for ws, we in windows)
        B = len(todo)
        # active region: the mel-rate stages and the phase need the whole window (their receptive fields are long: F0-net,
        # smoother of the lifter selection), the stages from the WaveNet on only the frames around what is emitted.
        # It starts on an aligned frame (same Winograd pairing as offline), at least `lead` frames inside a window that
        # does not start the utterance (pulses are reproducible from there), and is the same offset for every item.
        a0 = None
        for (sid, st, nn), (ws, we) in zip(todo, windows):
            ab = max(0, ((st.emitted - self.act_left) // self.align) * self.align) - ws
            if ws > 0 and ab < self.lead:
                ab = 0
            a0 = ab if a0 is None else min(a0, ab)
        if any(ws > 0 for ws, _ in windows) and 0 < a0 < self.lead:
            a0 = 0
        act = np.zeros((B,), dtype=np.int32)
        ends = []
        for bb, ((sid, st, nn), (ws, we)) in enumerate(zip(todo, windows)):
            a1 = we if we - (st.emitted + nn) <= self.act_right else st.emitted + nn + self.act_right
            ends.append(a1)
            act[bb] = a1 - ws - a0
        # carried sub-bands: usable when every item of the tick has a valid store and the same geometry
        sl, sr, wr = self.sr_left, self.sr_right, self.wn_reach
        wn = None
        desc = np.zeros((B, 5), dtype=np.int32)
        use_carry = self.carry
        sa = wa = None
        for (sid, st, nn), (ws, we) in zip(todo, windows):
            if not use_carry:
                break
            ok = st.carry_pos == st.emitted and st.emitted - sl >= ws
            sab = st.emitted - sl - ws
            wab = ((st.emitted + min(sr, st.carry_frames - sl) - wr) // self.align) * self.align - ws
            ok = ok and wab >= sab and (ws == 0 or wab >= self.lead) and st.carry_frames > sl
            ok = ok and (sa is None or (sab == sa and wab == wa))
            sa, wa = sab, wab
            use_carry = ok
        if use_carry and todo:
            a0 = sa
            wn = np.zeros((B,), dtype=np.int32)
            for bb, ((sid, st, nn), (ws, we)) in enumerate(zip(todo, windows)):
                act[bb] = ends[bb] - ws - sa
                wn[bb] = ends[bb] - ws - wa
                desc[bb, 1:3] = sa * spf, st.carry_frames * spf
        # rows the next tick will need: frames [e' - sr_left, e' + sr_right) around the next emit position e', if this tick
        # computes them exactly (its region reaches wn_reach frames beyond them, or to the end of the utterance)
        next_carry = []
        for bb, ((sid, st, nn), (ws, we)) in enumerate(zip(todo, windows)):
            e1 = st.emitted + nn
            lo, hi = e1 - sl, min(e1 + sr, ends[bb] if ends[bb] == we and st.closed else ends[bb] - wr)
            region_lo = ws + a0 if use_carry else ws + a0 + (wr if ws + a0 > 0 else 0)
            good = self.carry and lo >= region_lo and hi > lo + sl and lo >= ws
            desc[bb, 0] = st.slot
            if good:
                desc[bb, 3:5] = (lo - ws) * spf, (hi - lo) * spf
            next_carry.append((e1, hi - lo) if good else (None, 0))
        # per-layer WaveNet state.  Steady tick: every item has the state of a region that ended layer_rows rows in front
        # of this tick's region end, the sub-bands up to the WaveNet's reach in front of that, the same geometry inside
        # its window, and does not end its utterance here.  Any other tick runs the whole region and stores the state.
        ldesc = np.full((B, 3), -1, dtype=np.int32)
        layer_rows = 0
        next_layer_end = [None] * B
        if self.layer_carry:
            steady = bool(use_carry)
            geom = None
            for bb, ((sid, st, nn), (ws, we)) in enumerate(zip(todo, windows)):
                final = st.closed and ends[bb] >= we
                ldesc[bb, 0] = st.slot
                if st.layer_end is None or final or st.layer_end >= ends[bb]:
                    steady = False
                elif steady:
                    gg = ((ends[bb] - st.layer_end) * spf, ends[bb] - ws)
                    steady = (gg[0] >= self._layer_min_rows and (geom is None or gg == geom) and st.layer_end - wr >= ws + sa and
                              st.emitted - sl + st.carry_frames == st.layer_end - wr)
                    geom = gg
            region0 = (wa if use_carry else a0)
            for bb, ((sid, st, nn), (ws, we)) in enumerate(zip(todo, windows)):
                final = st.closed and ends[bb] >= we
                # state of this tick's region: exact if the region starts the utterance or reaches 2 * reach + 1 frames back
                long_enough = ws + region0 == 0 or ends[bb] - ws - region0 >= 3 * wr + 1
                if not final and (steady or long_enough):
                    ldesc[bb, 2] = (ends[bb] - ws) * spf
                    next_layer_end[bb] = ends[bb]
            if steady:
                layer_rows = geom[0]
                wa = geom[1] - layer_rows // spf - wr                 # frame of the first new sub-band row
                for bb, ((sid, st, nn), (ws, we)) in enumerate(zip(todo, windows)):
                    ldesc[bb, 1] = (st.layer_end - ws) * spf
                    wn[bb] = ends[bb] - ws - wa
        tpad = max(tmax, self._tcap)              # periodic schedules keep one window size for all phases (n_frames masks)
        mel = np.zeros((B, tpad, self.dims.mel_channels), dtype=np.float32)
        noise = np.zeros((B, tpad * spf), dtype=np.float32)
        nfr = np.zeros((B,), dtype=np.int32)
        states = np.zeros((B, 6), dtype=np.int32)
        st_f = np.zeros((B, 2), dtype=np.float32)
        next_state_frame = []
        for bb, ((sid, st, nn), (ws, we)) in enumerate(zip(todo, windows)):
            mel[bb, :we - ws] = self._in_mel[st.slot, ws - st.base:we - st.base]
            if self.dims.noise_sigma:
                noise[bb, :(we - ws) * spf] = self._in_noise[st.slot, ws - st.base:we - st.base].reshape(-1)
            nfr[bb] = we - ws
            # the carried state sits at frame st.state_frame (>= ws + lead, or 0 at the utterance start): pulses are
            # reproducible from there on.  The next state is captured where the NEXT window's reproducible region
            # starts: `lag` = left - lead frames in front of the next emit position.
            nxt = max(st.state_frame, st.emitted + nn - (self.left - self.lead))
            st_f[bb, 0], st_f[bb, 1] = st.state[0], st.state[1]
            states[bb, 2:5] = st.state[2], (st.state_frame - ws) * ppf, (nxt - ws) * ppf if nxt < we else -1
            next_state_frame.append(nxt)
        states[:, :2] = st_f.view(np.int32)                   # one mbx_stream_state per item (pack_state)
        dev = self.engine.device
        mel_d = torch.as_tensor(mel, device=dev)
        noise_d = torch.as_tensor(noise, device=dev) if self.dims.noise_sigma else None
        # every int32 argument of the call in one upload
        use_fe = self.fe_carry and self.carry and tpad <= self._fe_ring
        fpos = np.asarray([ws % self._fe_ring for ws, _ in windows], dtype=np.int32)
        parts = [states.ravel(), desc.ravel(), ldesc.ravel(), nfr, act, wn if wn is not None else act, fpos]
        ints_d = torch.as_tensor(np.concatenate(parts), device=dev)
        cuts = np.cumsum([0] + [pp.size for pp in parts])
        states_d, desc_d, ldesc_d, nfr_d, act_d, wn_d, fpos_d = (ints_d[cuts[ii]:cuts[ii + 1]] for ii in range(7))
        states_d, desc_d, ldesc_d = states_d.view(B, 6), desc_d.view(B, 5), ldesc_d.view(B, 3)
        self.last_tick_frames = int(nfr.sum())
        if self.time_device:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        self.last_tick_active_frames = int(act.sum())
        self.last_tick_wavenet_frames = int(wn.sum()) if wn is not None else int(act.sum())
        self.last_tick_layer_rows = layer_rows
        if layer_rows:
            self.last_tick_wavenet_frames = B * layer_rows // spf
        audio, state_out = self.engine.forward(
            mel_d, n_frames=nfr_d, noise=noise_d, stream_state=states_d,
            active=(a0, act_d, int(act.max())),
            wavenet=(wa, wn_d, int(wn.max())) if wn is not None else None,
            carry=(self._store, desc_d) if self.carry else None,
            layers=(self._layer_store, ldesc_d, layer_rows) if self.layer_carry else None,
            frontend=(self._fe_store, fpos_d, 0, 0) if use_fe else None)      # whole window computed, every frame kept
        if self.time_device:
            ev1.record()
            ev1.synchronize()
            self.last_tick_device_ms = ev0.elapsed_time(ev1)
        # only the emitted samples come back: the columns [lo, hi) of the window that hold some item's chunk
        lo = min((st.emitted - ws) * hop for (sid, st, nn), (ws, we) in zip(todo, windows))
        hi = max((st.emitted - ws + nn) * hop for (sid, st, nn), (ws, we) in zip(todo, windows))
        audio = audio[:, lo:hi].cpu().numpy()
        state_out = state_out.cpu().numpy()
        steady_ctx = None
        # (mbx_window_advance keeps the part of a window that stays in LDS: at most 64 KB per item -- wide windows, e.g. the
        # RMS normalisation with several smoothing iterations or deep pre-conditioning chains, run launch by launch)
        window_fits = tpad * max(self.dims.mel_channels, spf) * 4 <= 64 * 1024
        phases = {st.ticks % len(self.schedule) for _, st, _ in todo}
        if (layer_rows and self.use_graph and self.periodic and window_fits and len(phases) == 1 and
                len({nn for _, _, nn in todo}) == 1 and len(set(windows)) >= 1 and
                len({(st.emitted - ws, we - ws) for (_, st, _), (ws, we) in zip(todo, windows)}) == 1):
            # a steady tick: when this phase of the schedule comes round again with every stream continuing the same way,
            # it is this launch sequence on windows that moved on by one period -- every window-relative argument is the
            # same (_steady_status checks it)
            (sid0, st0, nn0), (ws0, we0) = todo[0], windows[0]
            steady_ctx = {
                "phase": phases.pop(), "chunk": nn0, "T": we0 - ws0, "tpad": tpad, "a0": a0, "wa": wa, "lo": lo, "hi": hi,
                "layer_rows": layer_rows, "act": act.copy(), "wn": wn.copy(), "nfr": nfr.copy(), "desc": desc.copy(),
                "ldesc": ldesc.copy(), "state_consts": states[:, 3:5].copy(), "rel0": st0.emitted - ws0, "use_fe": use_fe,
                "frames": self.last_tick_frames, "active_frames": self.last_tick_active_frames,
                "wavenet_frames": self.last_tick_wavenet_frames, "ws0": ws0}
        result = {}
        for bb, ((sid, st, nn), (ws, we)) in enumerate(zip(todo, windows)):
            a0 = (st.emitted - ws) * hop - lo
            result[sid] = audio[bb, a0:a0 + nn * hop].copy()
            st.emitted += nn
            st.ticks += 1
            st.carry_pos, st.carry_frames = next_carry[bb]
            st.layer_end = next_layer_end[bb]
            if next_state_frame[bb] < we:
                ff = state_out[bb, :2].copy().view(np.float32)
                st.state = (float(ff[0]), float(ff[1]), int(state_out[bb, 2]))
                st.state_frame = next_state_frame[bb]
        sids = [sid for sid, _, _ in todo]
        if steady_ctx is None:
            self._steady = None                   # not a steady tick: whatever run was being recorded is over
        else:
            run = self._steady
            if run is None or run["sids"] != sids or run["tpad"] != tpad:
                run = {"sids": sids, "streams": [st for _, st, _ in todo], "B": B, "tpad": tpad, "phases": {}, "graphs": {},
                       "slots": np.asarray([st.slot for _, st, _ in todo], dtype=np.int64), "win": None, "shared": None}
                self._steady = run
            old = run["phases"].get(steady_ctx["phase"])
            if old is not None and any(not np.array_equal(old[kk], steady_ctx[kk]) if isinstance(old[kk], np.ndarray)
                                       else old[kk] != steady_ctx[kk] for kk in steady_ctx if kk != "ws0"):
                run["graphs"].pop(steady_ctx["phase"], None)          # the geometry of this phase changed: capture it anew
            run["phases"][steady_ctx["phase"]] = steady_ctx
            # where the run stands: the windows this tick worked on (device tensors) and their first frame, the streams'
            # progress and phase states as vectors (replayed ticks keep them there, _sync_streams writes them back)
            run["win_src"] = (mel_d, noise_d)
            run["win_ws0"] = steady_ctx["ws0"]
            run["win_T"] = steady_ctx["T"]
            run["state_v"] = state_out.copy()
            run["emitted_v"] = np.asarray([st.emitted for _, st, _ in todo], dtype=np.int64)
            run["next_phase"] = todo[0][1].ticks % len(self.schedule)
            run["pending"], run["pending_frames"] = 0, 0
            run.pop("need_v", None)
        return result

    # ------------------------------------------------------------------------------------------------------------------
    # steady ticks as replayed hipGraphs (one per phase of the tick schedule)
    # ------------------------------------------------------------------------------------------------------------------
    def _sync_streams(self):
        """Write the progress of the replayed ticks back to the stream objects (inside a run of replayed ticks it is kept
        in vectors: emitted frames, carried phase states)."""
        run = self._steady
        if run is None or not run["pending"]:
            return
        adv = run["pending_frames"]
        sf = run["state_v"][:, :2].copy().view(np.float32)
        for bb, st in enumerate(run["streams"]):
            st.ticks += run["pending"]
            st.emitted += adv
            st.carry_pos = st.emitted
            st.layer_end += adv
            st.state_frame += adv
            st.state = (float(sf[bb, 0]), float(sf[bb, 1]), int(run["state_v"][bb, 2]))
        run["pending"], run["pending_frames"] = 0, 0

    def _leave_steady(self):
        self._sync_streams()
        self._steady = None

    def _steady_status(self):
        """What the tick about to run is for the steady run in self._steady: "replay" -- its phase of the schedule has been
        recorded and this tick is that recorded tick moved on (the same streams and no other one ready, each with a whole
        chunk and its look-ahead available and no utterance end inside the window, the same position of the window
        relative to the emitted frames: then every window-relative argument of the engine call is unchanged); "record" --
        the streams continue but this phase has no recorded tick yet (it runs launch by launch and is recorded);
        "broken" -- anything else."""
        run = self._steady
        streams, B = run["streams"], run["B"]
        if self._inputs_changed or "need_v" not in run:
            # frames received (+ whether the stream is closed: its last frame must then lie behind the window) and the
            # column of absolute frame 0 in the shared input rows; only push() changes them
            run["need_v"] = np.fromiter((st.have - st.closed for st in streams), np.int64, B)
            run["base_v"] = np.fromiter((st.base for st in streams), np.int64, B)
            self._inputs_changed = False
        phase = run["next_phase"]
        chunk = self.schedule[phase]
        emitted = run["emitted_v"]
        if int((run["need_v"] - emitted).min()) < chunk + self.right:
            return "broken"
        if len(self.streams) != B:                        # a stream outside the recorded set must not be ready
            inside = set(run["sids"])
            if any(self._ready(st) > 0 for sid, st in self.streams.items() if sid not in inside):
                return "broken"
        ctx = run["phases"].get(phase)
        if ctx is None or (phase - 1) % len(self.schedule) not in run["phases"]:
            return "record"                               # (a phase's graph shifts the window of the phase before it)
        ws = np.maximum(0, ((emitted - self.left) // self.align) * self.align)
        if np.any(emitted - ws != ctx["rel0"]) or ctx["chunk"] != chunk:
            return "broken"
        return "replay"

    def _capture(self, run, phase):
        """Fixed buffers + the captured launch sequence of the steady tick of one phase of the schedule.  All phases of a
        run work on ONE pair of device-resident windows (run["win"]: mel (B, tpad, channels), noise (B, tpad * spf)); a
        tick moves them on -- shift by the frames the aligned window start advanced since the tick before, append the
        tick's new frames -- and runs the forward pass with the phase's constant arguments."""
        import torch
        ctx = run["phases"][phase]
        n_ph = len(self.schedule)
        prev = run["phases"][(phase - 1) % n_ph]
        eng, dims, dev = self.engine, self.dims, self.engine.device
        B, tpad, chunk, spf, hop = run["B"], run["tpad"], ctx["chunk"], dims.steps_per_frame, dims.hop_size
        # the window of the tick before held T_prev frames from ws_prev on; this one T frames from ws on: the start moved
        # by `shift` = (emitted - rel0) - (emitted_prev - rel0_prev) frames, `chunk` frames are new at the end
        shift = prev["chunk"] - (ctx["rel0"] - prev["rel0"])
        keep = prev["T"] - shift
        if shift < 0 or keep < 0 or keep + chunk != ctx["T"] or ctx["T"] > tpad:
            raise RuntimeError(f"steady ticks of phases {(phase - 1) % n_ph} and {phase} do not chain: shift {shift}, "
                               f"windows {prev['T']} -> {ctx['T']} frames, chunk {chunk}")
        use_noise = bool(dims.noise_sigma)
        if run["win"] is None:
            mel_win = torch.zeros((B, tpad, dims.mel_channels), dtype=torch.float32, device=dev)
            noise_win = torch.zeros((B, tpad * spf), dtype=torch.float32, device=dev) if use_noise else None
            run["win"] = (mel_win, noise_win)
            run["shared"] = {
                "audio_buf": torch.empty((B, tpad * hop), dtype=torch.float32, device=dev),
                "state_out": torch.empty((B, 6), dtype=torch.int32, device=dev),
                "state_host": torch.empty((B, 6), dtype=torch.int32).pin_memory()}
        mel_win, noise_win = run["win"]
        shared = run["shared"]
        n_mel, n_noise = B * chunk * dims.mel_channels, (B * chunk * spf if use_noise else 0)
        # one pinned host buffer / one device buffer for everything a tick uploads: new mel frames, new noise, phase states
        stage_host = torch.empty(n_mel + n_noise + B * 7, dtype=torch.float32).pin_memory()
        stage_dev = torch.empty_like(stage_host, device=dev)
        mel_new = stage_dev[:n_mel].view(B, chunk, dims.mel_channels)
        noise_new = stage_dev[n_mel:n_mel + n_noise].view(B, chunk * spf) if use_noise else None
        states_d = stage_dev[n_mel + n_noise:n_mel + n_noise + B * 6].view(torch.int32).view(B, 6)
        fpos_d = stage_dev[n_mel + n_noise + B * 6:].view(torch.int32)
        # front end: the window gained `chunk` frames and the last fe_right frames of the window before were inexact: the
        # sub-nets run on the last chunk + fe_right (+ their reach) frames in front of the window's end (frame ctx["T"] of the
        # buffer: fe_end_frames), everything in front of that comes from the ring
        fe_new, fe_margin = chunk + self.fe_right, self.fe_left
        use_fe = ctx["use_fe"] and fe_new + fe_margin <= ctx["T"]
        ints = {kk: torch.as_tensor(ctx[kk], device=dev) for kk in ("act", "wn", "nfr", "desc", "ldesc")}
        audio_buf, state_out, state_host = shared["audio_buf"], shared["state_out"], shared["state_host"]
        audio_host = torch.empty((B, ctx["hi"] - ctx["lo"]), dtype=torch.float32).pin_memory()
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize(dev)
        # thread_local: another thread of the process (the RCCL watchdog of a multi-rank job) may touch the runtime while
        # this one captures
        with torch.cuda.graph(graph, capture_error_mode="thread_local"):
            stage_dev.copy_(stage_host, non_blocking=True)
            # shift the kept frames to the front and append the tick's new ones: one launch whatever the schedule
            eng.window_update(mel_win, mel_new, noise_win, noise_new, shift, keep)
            eng.forward(mel_win, n_frames=ints["nfr"], noise=noise_win, stream_state=states_d,
                        active=(ctx["a0"], ints["act"], int(ctx["act"].max())),
                        wavenet=(ctx["wa"], ints["wn"], int(ctx["wn"].max())),
                        carry=(self._store, ints["desc"].view(B, 5)) if self.carry else None,
                        layers=(self._layer_store, ints["ldesc"].view(B, 3), ctx["layer_rows"]),
                        frontend=(self._fe_store, fpos_d, fe_new if use_fe else 0, fe_margin if use_fe else 0, ctx["T"])
                        if ctx["use_fe"] else None,
                        out=audio_buf, state_out=state_out)
            # the emitted samples of every stream: one strided device-to-host copy (no device-side gather in between)
            eng.emit_rows(audio_buf, ctx["lo"], ctx["hi"] - ctx["lo"], audio_host)
            state_host.copy_(state_out, non_blocking=True)
        n_state = n_mel + n_noise
        return {"graph": graph, "stage_host": stage_host, "stage_np": stage_host.numpy(), "n_mel": n_mel,
                "arange": np.arange(chunk, dtype=np.int64), "n_noise": n_noise, "n_state": n_state,
                "audio_host": audio_host, "state_host": state_host, "shift": shift, "keep": keep,
                "keep_alive": (stage_dev, ints)}

    def _graph_tick(self):
        """A steady tick as one graph launch: gather and upload the new frames and the phase states, replay, read the
        chunk back.  Nothing here loops over the streams."""
        import torch
        run = self._steady
        phase = run["next_phase"]
        ctx = run["phases"][phase]
        n_ph = len(self.schedule)
        gg = run["graphs"].get(phase)
        if gg is None:
            try:
                if (phase - 1) % n_ph not in run["phases"]:
                    raise RuntimeError("the phase in front of this one has no recorded tick")
                gg = run["graphs"][phase] = self._capture(run, phase)
            except Exception as exc:                          # noqa: BLE001 -- whatever made the capture fail
                # the streams must not get stuck retrying a capture that cannot succeed: from here on every tick runs
                # launch by launch (bit-identical results, more host time)
                import sys
                print(f"mbexwn_vocoder_amd.streaming: hipGraph capture of the steady tick failed ({type(exc).__name__}: "
                      f"{exc}); ticks run launch by launch from here on", file=sys.stderr)
                self.use_graph = False
                self._leave_steady()
                return self.tick()
        dims, chunk = self.dims, ctx["chunk"]
        hop, B = dims.hop_size, run["B"]
        # the device windows hold the window of the tick before: written there by a replayed tick, or still in the tensors a
        # launch-by-launch tick uploaded
        if run["win_src"] is not run["win"]:
            src_mel, src_noise = run["win_src"]
            run["win"][0].copy_(src_mel)
            if run["win"][1] is not None:
                run["win"][1].copy_(src_noise)
            run["win_src"] = run["win"]
        stage = gg["stage_np"]
        emitted, slots = run["emitted_v"], run["slots"]
        base = run["base_v"]
        # the frames the windows gain: [emitted + right, emitted + right + chunk) of every stream
        cols = (emitted + self.right - base)[:, None] + gg["arange"]
        np.copyto(stage[:gg["n_mel"]].reshape(B, chunk, dims.mel_channels), self._in_mel[slots[:, None], cols])
        if gg["n_noise"]:
            np.copyto(stage[gg["n_mel"]:gg["n_state"]].reshape(B, chunk, dims.steps_per_frame),
                      self._in_noise[slots[:, None], cols])
        ints = stage[gg["n_state"]:].view(np.int32)
        states = ints[:B * 6].reshape(B, 6)
        ints[B * 6:] = (emitted - ctx["rel0"]) % self._fe_ring              # ring frame of each window's first frame
        states[:, :3] = run["state_v"][:, :3]
        states[:, 3:5] = ctx["state_consts"]
        states[:, 5] = 0
        if self.time_device:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
        gg["graph"].replay()
        if self.time_device:
            ev1.record()
        torch.cuda.current_stream(self.engine.device).synchronize()
        if self.time_device:
            self.last_tick_device_ms = ev0.elapsed_time(ev1)
        audio = gg["audio_host"].numpy().copy()
        run["state_v"] = gg["state_host"].numpy().copy()
        a0 = ctx["rel0"] * hop - ctx["lo"]
        result = dict(zip(run["sids"], audio[:, a0:a0 + chunk * hop]))
        emitted += chunk
        run["pending"] += 1
        run["pending_frames"] += chunk
        run["next_phase"] = (phase + 1) % n_ph
        self.last_tick_frames, self.last_tick_active_frames = ctx["frames"], ctx["active_frames"]
        self.last_tick_wavenet_frames, self.last_tick_layer_rows = ctx["wavenet_frames"], ctx["layer_rows"]
        self.last_tick_replayed = True
        self.graph_ticks += 1
        return result

    def finished(self, stream_id):
        self._sync_streams()
        st = self.streams[stream_id]
        return st.closed and st.emitted >= st.have
