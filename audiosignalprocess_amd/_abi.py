"""ctypes mirror of include/asp_ns.h (plain data layouts only)."""
import ctypes as C

import numpy as np

BLOCKL = 160
ANAL = 256
BINS = 129
SIMULT = 3
HIST = 1000

MEM_HOST = 0
MEM_DEVICE = 1


class AspNsState(C.Structure):
    """include/asp_ns.h: AspNsState (canonical per-stream snapshot)."""

    _fields_ = [
        ("fs", C.c_int32),
        ("aggrMode", C.c_int32),
        ("initFlag", C.c_int32),
        ("gainmap", C.c_int32),
        ("blockInd", C.c_int32),
        ("updates", C.c_int32),
        ("counter", C.c_int32 * SIMULT),
        ("modelUpdatePars", C.c_int32 * 4),
        ("overdrive", C.c_float),
        ("denoiseBound", C.c_float),
        ("priorSpeechProb", C.c_float),
        ("signalEnergy", C.c_float),
        ("sumMagn", C.c_float),
        ("whiteNoiseLevel", C.c_float),
        ("pinkNoiseNumerator", C.c_float),
        ("pinkNoiseExp", C.c_float),
        ("priorModelPars", C.c_float * 7),
        ("featureData", C.c_float * 7),
        ("analyzeBuf", C.c_float * ANAL),
        ("dataBuf", C.c_float * ANAL),
        ("syntBuf", C.c_float * ANAL),
        ("density", C.c_float * (SIMULT * BINS)),
        ("lquantile", C.c_float * (SIMULT * BINS)),
        ("quantile", C.c_float * BINS),
        ("smooth", C.c_float * BINS),
        ("noise", C.c_float * BINS),
        ("noisePrev", C.c_float * BINS),
        ("magnPrevAnalyze", C.c_float * BINS),
        ("magnPrevProcess", C.c_float * BINS),
        ("logLrtTimeAvg", C.c_float * BINS),
        ("magnAvgPause", C.c_float * BINS),
        ("initMagnEst", C.c_float * BINS),
        ("parametricNoise", C.c_float * BINS),
        ("speechProb", C.c_float * BINS),
        ("histLrt", C.c_int32 * HIST),
        ("histSpecFlat", C.c_int32 * HIST),
        ("histSpecDiff", C.c_int32 * HIST),
    ]

    def field_array(self, name):
        """numpy copy of one field (scalars become 0-d arrays)."""
        v = getattr(self, name)
        if isinstance(v, (int, float)):
            return np.asarray(v)
        return np.ctypeslib.as_array(v).copy()

    def to_dict(self):
        return {n: self.field_array(n) for n, _ in self._fields_}

    @classmethod
    def from_dict(cls, d):
        s = cls()
        for n, t in cls._fields_:
            v = d[n]
            if hasattr(t, "_length_"):
                arr = np.ctypeslib.as_array(getattr(s, n))
                arr[...] = np.asarray(v).reshape(arr.shape)
            else:
                setattr(s, n, np.asarray(v).item())
        return s


STATE_FIELDS = [n for n, _ in AspNsState._fields_]


class AspNsHbState(C.Structure):
    """include/asp_ns.h: AspNsHbState (high-band analysis buffers, ns_core.h:112)."""

    _fields_ = [("dataBufHB", C.c_float * (2 * 256))]


class AspBtState(C.Structure):
    """include/asp_bt.h: AspBtState (carried state between macroblocks)."""

    _fields_ = [("win_size", C.c_int32), ("inbuf_tail", C.c_float * 1024), ("out_tail", C.c_float * 1024)]


class AecConfig(C.Structure):
    """include/asp_aec.h: AecConfig (echo_cancellation.h:37-43)."""

    _fields_ = [("nlpMode", C.c_int16), ("skewMode", C.c_int16), ("metricsMode", C.c_int16),
                ("delay_logging", C.c_int)]


class AspAecState(C.Structure):
    """include/asp_aec.h: AspAecState (per-stream float state of AecCore)."""

    _fields_ = [
        ("dBuf", C.c_float * 128), ("eBuf", C.c_float * 128),
        ("xPow", C.c_float * 65), ("dPow", C.c_float * 65), ("dMinPow", C.c_float * 65),
        ("dInitMinPow", C.c_float * 65),
        ("xfBuf", C.c_float * (2 * 32 * 65)), ("wfBuf", C.c_float * (2 * 32 * 65)),
        ("sde", C.c_float * (65 * 2)), ("sxd", C.c_float * (65 * 2)),
        ("xfwBuf", C.c_float * (32 * 65 * 2)),
        ("sx", C.c_float * 65), ("sd", C.c_float * 65), ("se", C.c_float * 65),
        ("outBuf", C.c_float * 64),
        ("hNlFbMin", C.c_float), ("hNlFbLocalMin", C.c_float), ("hNlXdAvgMin", C.c_float),
        ("overDrive", C.c_float), ("overDriveSm", C.c_float),
        ("hNlNewMin", C.c_int32), ("hNlMinCtr", C.c_int32),
        ("delayIdx", C.c_int32), ("stNearState", C.c_int32), ("echoState", C.c_int32),
        ("divergeState", C.c_int32),
        ("xfBufBlockPos", C.c_int32), ("noiseEstCtr", C.c_int32), ("delayEstCtr", C.c_int32),
        ("seed", C.c_uint32),
        ("dBufH", C.c_float * 128),
    ]


class AspAecPowerLevel(C.Structure):
    _fields_ = [("sfrsum", C.c_float), ("sfrcounter", C.c_int32), ("framelevel", C.c_float), ("frsum", C.c_float),
                ("frcounter", C.c_int32), ("minlevel", C.c_float), ("averagelevel", C.c_float)]


class AspAecStats(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("instant", "average", "min", "max", "sum", "hisum", "himean")] + \
               [("counter", C.c_int32), ("hicounter", C.c_int32)]


class AspAecMetricsState(C.Structure):
    """include/asp_aec.h: AspAecMetricsState."""

    _fields_ = [("farlevel", AspAecPowerLevel), ("nearlevel", AspAecPowerLevel), ("linoutlevel", AspAecPowerLevel),
                ("nlpoutlevel", AspAecPowerLevel), ("erl", AspAecStats), ("erle", AspAecStats), ("aNlp", AspAecStats),
                ("rerl", AspAecStats), ("stateCounter", C.c_int32)]

    def to_array(self):
        import numpy as np
        return np.frombuffer(bytes(self), dtype=np.uint32).copy()


class AecLevel(C.Structure):
    _fields_ = [("instant", C.c_int), ("average", C.c_int), ("max", C.c_int), ("min", C.c_int)]


class AecMetrics(C.Structure):
    _fields_ = [("rerl", AecLevel), ("erl", AecLevel), ("erle", AecLevel), ("aNlp", AecLevel)]

    def to_tuple(self):
        return tuple(getattr(getattr(self, a), f) for a in ("rerl", "erl", "erle", "aNlp")
                     for f in ("instant", "average", "max", "min"))


class AspAecControl(C.Structure):
    """include/asp_aec.h: AspAecControl (integer control plane shared by a batch)."""

    _fields_ = [(n, C.c_int32) for n in (
        "startup_phase", "checkBuffSize", "bufSizeStart", "knownDelay", "filtDelay",
        "timeForDelayChange", "lastDelayDiff", "counter", "sum", "firstVal", "checkBufSizeCtr",
        "system_delay", "core_knownDelay",
        "far_read", "far_write", "far_wrap", "pre_read", "pre_write", "pre_wrap",
        "near_read", "near_write", "near_wrap", "out_read", "out_write", "out_wrap",
        "blocks_processed")]


class AspAecDelayState(C.Structure):
    """include/asp_aec.h: AspAecDelayState (the delay estimator of one stream and the AecCore fields around it)."""

    _fields_ = [("mean_far_spectrum", C.c_float * 65), ("far_spectrum_initialized", C.c_int32),
                ("binary_far_history", C.c_uint32 * 125), ("far_bit_counts", C.c_int32 * 125),
                ("mean_near_spectrum", C.c_float * 65), ("near_spectrum_initialized", C.c_int32),
                ("binary_near_history", C.c_uint32 * 126), ("mean_bit_counts", C.c_int32 * 126),
                ("bit_counts", C.c_int32 * 125), ("histogram", C.c_float * 126),
                ("minimum_probability", C.c_int32), ("last_delay_probability", C.c_int32), ("last_delay", C.c_int32),
                ("last_candidate_delay", C.c_int32), ("compare_delay", C.c_int32), ("candidate_hits", C.c_int32),
                ("last_delay_histogram", C.c_float), ("lookahead", C.c_int32), ("allowed_offset", C.c_int32),
                ("delay_histogram", C.c_int32 * 125),
                ("previous_delay", C.c_int32), ("delay_correction_count", C.c_int32), ("shift_offset", C.c_int32),
                ("delay_quality_threshold", C.c_float),
                ("far_read", C.c_int32), ("far_write", C.c_int32), ("far_wrap", C.c_int32), ("system_delay", C.c_int32)]

    def far_available(self):
        """readable partitions of the stream's far buffer (ring_buffer.c:231-240); a reference export carries the
        count itself in far_read (far_write = -1)"""
        if self.far_write < 0:
            return self.far_read
        return self.far_write - self.far_read if self.far_wrap == 0 else 250 - self.far_read + self.far_write

    def diff(self, other, skip=("far_read", "far_write", "far_wrap")):
        """names of the fields that differ bitwise (ring positions compared as the readable count)"""
        import numpy as np
        bad = []
        for name, _t in self._fields_:
            if name in skip:
                continue
            a, b = getattr(self, name), getattr(other, name)
            if hasattr(a, "_length_"):
                if not np.array_equal(np.frombuffer(bytes(a), np.uint32), np.frombuffer(bytes(b), np.uint32)):
                    bad.append(name)
            elif isinstance(a, float):
                if np.float32(a).view(np.uint32) != np.float32(b).view(np.uint32):
                    bad.append(name)
            elif a != b:
                bad.append(name)
        if self.far_available() != other.far_available():
            bad.append("far_available")
        return bad


def aec_state_arrays(st):
    """AspAecState -> {field: numpy array / scalar} for comparisons."""
    import numpy as np

    out = {}
    for name, ctype in st._fields_:
        v = getattr(st, name)
        out[name] = np.ctypeslib.as_array(v).copy() if hasattr(v, "_length_") else v
    return out
