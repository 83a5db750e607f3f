for spec in "26 6 4 fp64" "28 4 2 fp64" "26 6 4 fp32" "28 6 2 fp32"; do
  for env in "QSV_X=0" "QSV_LOW_BITS=3 QSV_LANE_BITS=3" "QSV_TILE_BITS=12 QSV_REG_BITS=3" "QSV_LOW_BITS=3 QSV_LANE_BITS=3 QSV_TILE_BITS=12 QSV_REG_BITS=3" "QSV_LOW_BITS=3 QSV_LANE_BITS=3 QSV_TILE_BITS=12 QSV_REG_BITS=4" "QSV_LOW_BITS=4 QSV_LANE_BITS=4"; do
    echo -n "$spec | $env | "; env $env timeout -k 10 200 python scripts/big_n_geometry.py $spec 2>/dev/null | tail -1
  done
done
