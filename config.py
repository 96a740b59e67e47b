"""Configuration surface kept from the reference (reference config.py:9-136): the same dataclass names and
fields, the same ENVIRONMENT profiles, so scripts that do ``from config import get_config`` keep working.
Only ``interpolation.method`` and ``interpolation.min_data_points`` reach the hot path (reference
batch_processor.py:21-24).  ``python-dotenv`` is optional here (absent in the build image): a missing
package just means no .env file is read.  ``data_dir`` is this build's DB-less frame store location."""
import os
from dataclasses import dataclass, field

try:  # the reference hard-requires python-dotenv; treat it as optional
    from dotenv import load_dotenv
    load_dotenv()
except ImportError:
    pass


@dataclass
class DatabaseConfig:
    host: str = os.getenv("DB_HOST", "localhost")
    database: str = os.getenv("DB_DATABASE", "trading_data")
    user: str = os.getenv("DB_USER", "postgres")
    password: str = os.getenv("DB_PASSWORD", "")
    port: int = int(os.getenv("DB_PORT", "5432"))

    def to_dict(self):
        return {"host": self.host, "database": self.database, "user": self.user,
                "password": self.password, "port": self.port}


@dataclass
class ProcessingConfig:
    max_workers: int = 32
    symbols_per_batch: int = 100        # here: symbols per device launch (interpolate_batch)
    chunk_size: int = 50000
    memory_limit_gb: int = 16
    enable_logging: bool = True
    log_level: str = "INFO"


@dataclass
class InterpolationConfig:
    frequency: str = "1min"
    method: str = "linear"
    max_gap_hours: int = 48
    min_data_points: int = 10
    extrapolate: bool = False
    preserve_greeks: bool = True


@dataclass
class CandleReconstructionConfig:
    target_frequency: str = "5min"
    source_frequency: str = "1min"
    min_candles_required: int = 5
    validate_ohlc: bool = True
    batch_size: int = 1000


@dataclass
class DataBridgeConfig:
    conversion_strategy: str = "spread_simulation"
    spread_method: str = "adaptive"
    enable_quality_checks: bool = True
    spread_parameters: dict = None

    def __post_init__(self):
        if self.spread_parameters is None:
            self.spread_parameters = {"base_spread_percent": 0.002, "volatility_factor": 1.5,
                                      "min_spread_percent": 0.0005, "max_spread_percent": 0.02,
                                      "trend_strength": 0.6}


@dataclass
class Config:
    database: DatabaseConfig
    processing: ProcessingConfig
    interpolation: InterpolationConfig
    candle_reconstruction: CandleReconstructionConfig
    data_bridge: DataBridgeConfig
    output_dir: str = "./interpolated_data"
    log_dir: str = "./logs"
    environment: str = "production"
    debug: bool = False
    data_dir: str = field(default_factory=lambda: os.getenv("IVS_DATA_DIR", "./frame_store"))


_PROFILES = {
    "development": dict(max_workers=4, symbols_per_batch=10, chunk_size=1000, log_level="DEBUG"),
    "testing": dict(max_workers=8, symbols_per_batch=25, chunk_size=10000, log_level="INFO"),
    "production": dict(max_workers=32, symbols_per_batch=100, chunk_size=50000, log_level="INFO"),
}


def get_config() -> Config:
    env = os.getenv("ENVIRONMENT", "production")
    prof = _PROFILES.get(env, _PROFILES["production"])
    return Config(database=DatabaseConfig(), processing=ProcessingConfig(enable_logging=True, **prof),
                  interpolation=InterpolationConfig(), candle_reconstruction=CandleReconstructionConfig(),
                  data_bridge=DataBridgeConfig(), debug=(env == "development"), environment=env)
