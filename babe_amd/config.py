"""Plain-YAML config loader producing the attribute-style nested mapping the reference reads
(OmegaConf in the reference: test.py:69).  PyYAML parses exponent floats without a dot
('1e-4') as strings; they are coerced to float like OmegaConf does."""
import re

_FLOAT = re.compile(r"[+-]?(\d+\.?\d*|\.\d+)[eE][+-]?\d+")


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


def to_attr(o):
    if isinstance(o, dict):
        return AttrDict({k: to_attr(v) for k, v in o.items()})
    if isinstance(o, (list, tuple)):
        return [to_attr(v) for v in o]
    if isinstance(o, str) and _FLOAT.fullmatch(o):
        return float(o)
    return o


def load_yaml(path):
    import yaml
    with open(path) as f:
        return to_attr(yaml.safe_load(f))


def default_args(sample_rate=44100, audio_len=368368, Ns=(64, 96, 96, 128, 128, 256, 256), T=35, xi=0.2,
                 start_sigma=0.2):
    """The blind-BWE configuration the benchmark is quoted on: conf/tester/blind_bwe_formal_3000_opt_2.yaml,
    conf/network/cqtdiff+.yaml, conf/exp/maestro44k_8s.yaml, conf/diff_params/edm.yaml (values restated)."""
    return to_attr(dict(
        exp=dict(sample_rate=sample_rate, audio_len=audio_len),
        network=dict(use_fencoding=False, use_norm=True, emb_dim=256, Ns=list(Ns), Ss=[2] * 7,
                     num_dils=[2, 3, 4, 5, 6, 7, 7], attention_layers=[0] * 8, attention_dict=None,
                     bottleneck_type="res_dil_convs", num_bottleneck_layers=1,
                     cqt=dict(window="kaiser", beta=1, num_octs=7, bins_per_oct=64)),
        diff_params=dict(sigma_data=0.063, sigma_min=1e-5, sigma_max=10, P_mean=-1.2, P_std=1.2, ro=13, ro_train=10,
                         Schurn=5, Snoise=1, Stmin=0, Stmax=50, aweighting=dict(use_aweighting=False, ntaps=101)),
        tester=dict(
            T=T, order=2, filter_out_cqt_DC_Nyq=True,
            posterior_sampling=dict(xi=xi, data_consistency=False, norm=2, smoothl1_beta=1, SNR_observations="None",
                                    start_sigma=start_sigma, freq_weighting="None", freq_weighting_filter="sqrt",
                                    stft_distance=dict(mag=False, logmag=False, use=False, use_multires=False, nfft=2048)),
            diff_params=dict(same_as_training=False, sigma_data=0.063, sigma_min=1e-4, sigma_max=1, ro=8, Schurn=10,
                             Snoise=1.0, Stmin=0, Stmax=50),
            blind_bwe=dict(NFFT=4096, fcmin=20, fcmax="nyquist", Amin=-50, Amax=30, sigma_den_estimate=0.0,
                           SNR_observations="None",
                           initial_conditions=dict(fc=[280, 285, 290, 295, 300], A=[-15, -17, -20, -25, -30]),
                           optimization=dict(max_iter=100, tol=[5e-3, 5e-3], mu=[1000, 10], clamp_fc=True,
                                             clamp_A=True, only_negative_A=True)),
            complete_recording=dict(inpaint_DC=True))))
